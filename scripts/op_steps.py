"""Run fwd+bwd steps of one hot-path op through the functional API (for rocprofv3 passes: kernel stats, PMC counters).
usage: python scripts/op_steps.py <compressor|expander|gain|distortion|parametric_eq> B C N [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D

op, B, C, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev, SR = "cuda", 44100
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
ctl = lambda lo, hi, n=B: (rnd(n) * (hi - lo) + lo).requires_grad_(True)
x = (rnd(B, C, N) * 2 - 1).requires_grad_(True)
w = torch.randn(B, C, N, device=dev, generator=g)
dyn = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]
peq = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
       (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
cs = {"compressor": [ctl(lo, hi) for lo, hi in dyn], "expander": [ctl(lo, hi) for lo, hi in dyn], "gain": [ctl(-24, 24)],
      "distortion": [ctl(0, 24, B * C)], "parametric_eq": [ctl(lo, hi) for lo, hi in peq]}[op]
fn = getattr(D, op)
for _ in range(steps):
    x.grad = None
    for c in cs:
        c.grad = None
    fn(x, SR, *cs).backward(w)
torch.cuda.synchronize()
print(op, B, C, N, steps, float(x.grad.abs().mean()))
