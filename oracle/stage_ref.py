"""Recipe for oracle/_ref: an importable archive of the reference package, built from the sources where they lie.

TEST / MEASUREMENT INFRASTRUCTURE ONLY. The reference (csteinmetz1/dasp-pytorch v0.0.1) is pure Python, so there is nothing to compile;
"building" it means packing /root/reference/dasp_pytorch/*.py into oracle/_ref/dasp_pytorch_ref.zip (zipimport makes that importable:
`sys.path.insert(0, ".../dasp_pytorch_ref.zip"); import dasp_pytorch`). oracle/_ref/ is git-ignored - no reference source enters the
history - but it is not gpurun-ignored, so the archive travels to the GPU box with the snapshot, where /root/reference does not exist.
Only bench.py's cpu_baseline leg (the reference timed on the host cores beside the GPU number) and tests may import it.

    python oracle/stage_ref.py            # or __graft_entry__.build(), which calls stage() when /root/reference is present
"""
import os
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PKG = "/root/reference/dasp_pytorch"
OUT = os.path.join(HERE, "_ref", "dasp_pytorch_ref.zip")


def stage(ref_pkg=REF_PKG, out=OUT):
    """Returns the archive path, or None when the reference checkout is not present (the GPU box: the prebuilt archive is used)."""
    if not os.path.isdir(ref_pkg):
        return None
    os.makedirs(os.path.dirname(out), exist_ok=True)
    files = sorted(f for f in os.listdir(ref_pkg) if f.endswith(".py"))
    with zipfile.ZipFile(out, "w", zipfile.ZIP_DEFLATED) as z:
        for f in files:
            info = zipfile.ZipInfo("dasp_pytorch/" + f, date_time=(2024, 1, 1, 0, 0, 0))     # fixed timestamps: reproducible archive
            info.compress_type = zipfile.ZIP_DEFLATED
            z.writestr(info, open(os.path.join(ref_pkg, f), "rb").read())
    return out


if __name__ == "__main__":
    print(stage())
