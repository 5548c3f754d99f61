"""Build libdasp_hip.so (gfx950) in-tree with hipcc. No torch dependency: the library is a plain
C-ABI shared object (include/dasp_hip.h) that the Python side binds with ctypes."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libdasp_hip.so")
ARCH = "gfx950"


def kernel_source_hash(files=("sosfilt.hip", "sos_tile.hpp", "common.hpp"), root=HERE):
    """sha256 (16 hex digits) over the code of the cascaded-biquad kernels - comments stripped, whitespace collapsed - so that off-line
    measurements (profiles/rNN/hbm_traffic.json) can be tied to the kernels they were taken from without breaking on a reworded comment."""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in files:
        src = open(os.path.join(root, f)).read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
        src = re.sub(r"//[^\n]*", " ", src)
        h.update(f.encode())
        h.update(" ".join(src.split()).encode())
    return h.hexdigest()[:16]


def sources():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hpp")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs, jobs = [], []
    for src in sources():
        obj = src[:-4] + ".o"
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(src), *[os.path.getmtime(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".hpp")]):
            jobs.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-pass-failed", "-Wno-inline-asm", "-c", src, "-o", obj])
        objs.append(obj)
    if jobs:        # one hipcc per stale source, side by side (sosfilt.hip alone is most of a serial build)
        from concurrent.futures import ThreadPoolExecutor
        if verbose:
            for cmd in jobs:
                print(" ".join(cmd), file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(subprocess.check_call, jobs))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
