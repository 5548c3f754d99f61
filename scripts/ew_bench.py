"""Developer timing of gain / distortion fwd+bwd at (256,2,131072)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import _lib
dev = "cuda"
x = (torch.rand(256, 2, 131072, device=dev) * 2 - 1).requires_grad_(True)
w = torch.randn(256, 2, 131072, device=dev)
ctl = {"gain": (torch.rand(256, device=dev) * 24).requires_grad_(True), "distortion": (torch.rand(256, 2, device=dev) * 24).requires_grad_(True)}
def step(fn, g):
    x.grad = None; g.grad = None
    fn(x, 44100, g).backward(w)
for name, fn in (("gain", D.gain), ("distortion", D.distortion)):
    g = ctl[name]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.7:
        for _ in range(20): step(fn, g)
        torch.cuda.synchronize()
    _lib.timers.start()
    for _ in range(100): step(fn, g)
    kt = _lib.timers.stop()
    print(name, {k: round(sum(v) / len(v), 4) for k, v in kt.items()})
