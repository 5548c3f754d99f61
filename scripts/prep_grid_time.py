"""Developer timing (run under rocprofv3 --kernel-trace): the design launch with and without the basis workgroups - 40 forward passes under
no_grad (no backward follows: 16 workgroups), then 40 forward + backward steps (32 workgroups). scripts/prep_grid_report.py reads the trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from bench import PEQ_RANGES, SR
B, C, N = 16, 2, 131072
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
cols = [(rnd(B) * (hi - lo) + lo).requires_grad_(True) for lo, hi in PEQ_RANGES]
x = (rnd(B, C, N) * 2 - 1).requires_grad_(True)
w = torch.randn(B, C, N, device=dev, generator=g)
for _ in range(40):
    with torch.no_grad():
        D.parametric_eq(x, SR, *cols)
torch.cuda.synchronize()
for _ in range(40):
    x.grad = None
    for c in cols: c.grad = None
    D.parametric_eq(x, SR, *cols).backward(w)
torch.cuda.synchronize()
