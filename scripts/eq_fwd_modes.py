"""Developer timing: parametric_eq forward with and without saving the chunk states (grad mode vs no_grad)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import _lib
dev = "cuda"
B, C, N = 256, 2, 131072
x = torch.rand(B, C, N, device=dev) * 2 - 1
R = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
     (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
cols = [torch.rand(B, device=dev) * (hi - lo) + lo for lo, hi in R]
for mode in ("no_grad", "grad"):
    xx = x.clone().requires_grad_(mode == "grad")
    def step():
        if mode == "grad":
            return D.parametric_eq(xx, 44100, *cols)
        with torch.no_grad():
            return D.parametric_eq(xx, 44100, *cols)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.7:
        for _ in range(20): step()
        torch.cuda.synchronize()
    _lib.timers.start()
    for _ in range(200): step()
    kt = _lib.timers.stop()
    print(mode, {k: round(sum(v) / len(v), 4) for k, v in kt.items() if "forward" in k})
