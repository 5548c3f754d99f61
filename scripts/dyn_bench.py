"""Developer timing of the compressor kernels (C ABI entry points) at BASELINE config 3."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import _lib
B, C, N = 256, 2, 262144
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(B, C, N, device=dev, generator=g) * 2 - 1).requires_grad_(True)
rng = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]
ctl = [(torch.rand(B, device=dev, generator=g) * (hi - lo) + lo).requires_grad_(True) for lo, hi in rng]
w = torch.randn(B, C, N, device=dev, generator=g)
def step():
    x.grad = None
    for c in ctl: c.grad = None
    D.compressor(x, 44100, *ctl).backward(w)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    for _ in range(10): step()
    torch.cuda.synchronize()
_lib.timers.start()
for _ in range(50): step()
kt = _lib.timers.stop()
for k, v in kt.items():
    print(k, f"{sum(v)/len(v):.4f} ms")
