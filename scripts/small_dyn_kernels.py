"""Developer timing: the kernels of a small-batch compressor step (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
B, C, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 2, 262144)))
R = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]
g = torch.Generator(device="cuda:0").manual_seed(0)
x = (torch.rand(B, C, N, device="cuda:0", generator=g) * 2 - 1).requires_grad_(True)
cols = [(torch.rand(B, device="cuda:0", generator=g) * (hi - lo) + lo).requires_grad_(True) for lo, hi in R]
w = torch.randn(B, C, N, device="cuda:0", generator=g)
for _ in range(60):
    x.grad = None
    for c in cols: c.grad = None
    D.compressor(x, 44100, *cols).backward(w)
torch.cuda.synchronize()
