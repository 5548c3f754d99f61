"""Build libdasp_hip.so (gfx950) in-tree with hipcc. No torch dependency: the library is a plain
C-ABI shared object (include/dasp_hip.h) that the Python side binds with ctypes."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libdasp_hip.so")
ARCH = "gfx950"


def kernel_source_hash(files=("sosfilt.hip", "sos_tile.hpp", "sos_gram_fin.hpp", "common.hpp"), root=HERE):
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


HEADER = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "dasp_hip.h")


def abi_hash(header=HEADER):
    """63-bit hash of include/dasp_hip.h (comments stripped, whitespace collapsed): compiled into libdasp_hip.so (csrc/abi.hip,
    dasp_abi_hash()) and into libdasp_torch.so (torch.ops.dasp._abi_hash()), compared when the extension is loaded."""
    import hashlib
    import re
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return int(hashlib.sha256(" ".join(src.split()).encode()).hexdigest()[:15], 16)


def sources():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hpp")] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs, jobs = [], []
    for src in sources():
        obj = src[:-4] + ".o"
        stamp = os.path.basename(src) == "abi.hip"          # the ABI stamp depends on the header and on nothing else
        newest = max(os.path.getmtime(src), os.path.getmtime(HEADER)) if stamp else max(
            os.path.getmtime(src), *[os.path.getmtime(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".hpp")])
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            jobs.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-pass-failed", "-Wno-inline-asm"]
                        + ([f"-DDASP_ABI_HASH={abi_hash()}ULL"] if stamp else []) + ["-c", src, "-o", obj])
        objs.append(obj)
    if jobs:        # one hipcc per stale source, side by side (sosfilt.hip alone is most of a serial build)
        from concurrent.futures import ThreadPoolExecutor
        if verbose:
            for cmd in jobs:
                print(" ".join(cmd), file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(subprocess.check_call, jobs))
    # (soname: the torch extension libdasp_torch.so names this library as a dependency; a variant build loaded first through DASP_HIP_LIB
    # is then the one the dynamic loader hands to it)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,-soname,libdasp_hip.so", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


TORCH_EXT_SRC = os.path.join(HERE, "torch_ext", "dasp_torch_ops.cpp")
TORCH_EXT = os.path.join(HERE, "libdasp_torch.so")


TORCH_EXT_OBJ = os.path.join(HERE, "torch_ext", "dasp_torch_ops.o")


def _torch_ext_cmds():
    import torch
    from torch.utils import cpp_extension
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = [f"-I{p}" for p in cpp_extension.include_paths()] + ["-I/opt/rocm/include", f"-I{os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include')}"]
    compile_cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
                   f"-DDASP_ABI_HASH={abi_hash()}LL"] + inc + ["-c", TORCH_EXT_SRC, "-o", TORCH_EXT_OBJ]
    link_cmd = ["g++", "-shared", "-fPIC", TORCH_EXT_OBJ, "-o", TORCH_EXT, f"-L{tlib}", "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip", f"-L{HERE}", "-ldasp_hip",
                "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    return compile_cmd, link_cmd


def _compile_torch_ext_obj(force=False, verbose=False):
    """The extension's one translation unit (needs the torch headers and include/dasp_hip.h only - not the kernel library)."""
    if not force and os.path.exists(TORCH_EXT_OBJ) and os.path.getmtime(TORCH_EXT_OBJ) >= max(os.path.getmtime(TORCH_EXT_SRC), os.path.getmtime(HEADER)):
        return False
    cmd = _torch_ext_cmds()[0]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return True


def build_torch_ext(force=False, verbose=False):
    """csrc/torch_ext/dasp_torch_ops.cpp -> csrc/libdasp_torch.so: the TORCH_LIBRARY registration of torch.ops.dasp.* over the C ABI.
    Host-only C++ (no kernels): compiled with g++ against the torch headers of the running interpreter and linked to libdasp_hip.so
    next to it ($ORIGIN). In-tree, like the kernel library, so that it travels to the GPU box and shows up among the loaded objects."""
    lib = build_lib()
    recompiled = _compile_torch_ext_obj(force, verbose)
    if not recompiled and os.path.exists(TORCH_EXT) and os.path.getmtime(TORCH_EXT) >= max(os.path.getmtime(TORCH_EXT_OBJ), os.path.getmtime(lib)):
        return TORCH_EXT
    cmd = _torch_ext_cmds()[1]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return TORCH_EXT


def build_all(force=False, verbose=False):
    """Kernel library and torch extension; the extension's g++ compile (~20 s of torch headers) runs beside the hipcc jobs."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1) as pool:
        obj = pool.submit(_compile_torch_ext_obj, force, verbose)
        lib = build_lib(force, verbose)
        obj.result()
    # the jump-ahead table of csrc/mtrand.hip (a few seconds of integer arithmetic; cached next to the library, rebuilt on first use if absent)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from dasp_pytorch_amd import _mt19937
    _mt19937.build_table(force=force)
    return lib, build_torch_ext(verbose=verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
