"""Developer timing: cycle stamps of the design launch (table workgroup 7, basis workgroup of item 0) for segmented rows. Needs a -DDASP_TRACE
library (scripts/build_variant_sos.sh trace -DDASP_TRACE) in DASP_HIP_LIB."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import _lib
from bench import PEQ_RANGES, SR
B, C, N = 16, 2, 131072
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
cols = [(rnd(B) * (hi - lo) + lo).requires_grad_(True) for lo, hi in PEQ_RANGES]
x = (rnd(B, C, N) * 2 - 1).requires_grad_(True)
lib = ctypes.CDLL(_lib.LIB_PATH)
tr = (ctypes.c_longlong * 64)()
rows = []
for it in range(10):
    y = D.parametric_eq(x, SR, *cols)
    torch.cuda.synchronize()
    lib.dasp_debug_trace(tr)
    t = list(tr)
    rows.append((t[41] - t[40], t[42] - t[41], max(t[43], t[44]) - t[42], t[45] - max(t[43], t[44]), t[45] - t[40], t[49] - t[48], t[50] - t[49]))
rows = rows[2:]
names = ("design", "phi", "chains", "tables", "table workgroup to the end of the tables", "basis workgroup: design", "basis responses")
print("design launch, cycles (median): " + "  ".join(f"{n} {sorted(r[i] for r in rows)[len(rows) // 2]}" for i, n in enumerate(names)))
