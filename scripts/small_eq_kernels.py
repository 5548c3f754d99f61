"""Developer timing: the kernels of a small-batch parametric_eq step (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
B, C, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 2, 131072)))
R = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
     (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
g = torch.Generator(device="cuda:0").manual_seed(0)
x = (torch.rand(B, C, N, device="cuda:0", generator=g) * 2 - 1).requires_grad_(True)
cols = [(torch.rand(B, device="cuda:0", generator=g) * (hi - lo) + lo).requires_grad_(True) for lo, hi in R]
w = torch.randn(B, C, N, device="cuda:0", generator=g)
for _ in range(60):
    x.grad = None
    for c in cols: c.grad = None
    D.parametric_eq(x, 44100, *cols).backward(w)
torch.cuda.synchronize()
