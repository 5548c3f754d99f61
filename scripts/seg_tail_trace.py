"""Developer timing: cycle stamps of the finalize tail of the segmented Gram pass (item 0's last workgroup). Needs a library built with
-DDASP_TRACE (scripts/build_variant_sos.sh trace -DDASP_TRACE) and DASP_HIP_LIB pointing at it. usage: seg_tail_trace.py [B C N]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import _lib
from bench import PEQ_RANGES, SR
B, C, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 2, 131072)))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
cols = [(rnd(B) * (hi - lo) + lo).requires_grad_(True) for lo, hi in PEQ_RANGES]
x = (rnd(B, C, N) * 2 - 1).requires_grad_(True)
w = torch.randn(B, C, N, device=dev, generator=g)
lib = ctypes.CDLL(_lib.LIB_PATH)
tr = (ctypes.c_longlong * 64)()
rows = []
for it in range(12):
    x.grad = None
    for c in cols:
        c.grad = None
    D.parametric_eq(x, SR, *cols).backward(w)
    torch.cuda.synchronize()
    lib.dasp_debug_trace(tr)
    t = list(tr)
    rows.append((t[58] - t[57], t[59] - t[58], t[60] - t[59], t[61] - t[60], t[61] - t[57]))
rows = rows[2:]
names = ("prefetch issue + reduce + scatter", "P = C FW + lag sums + store", "wait + hand-off", "sum over workgroups + emit", "tail total")
med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(5)]
print(f"({B},{C},{N}) finalize tail of the workgroup that finalized item 0, shader clock cycles (median of {len(rows)}): " + "  ".join(f"{n} {v}" for n, v in zip(names, med)))
