"""Developer timing of noise_shaped_reverberation fwd+bwd (GPU time of the library calls, HIP events) at a given shape:
python scripts/reverb_time.py [B C N]; DASP_REVERB_CHUNK=<signals per pass> overrides the planner (0 = all signals at once);
DASP_RV_NOISE=generated (default: device_noise=True, the noise generated inside the filter-bank kernels) | explicit (a resident noise tensor) |
randn (a fresh torch.randn on the device per call, the round-2 device_noise path)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import _lib
B, C, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 2, 262144)))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, C, N, device=dev, generator=g).mul_(2).sub_(1).requires_grad_(True)
ctl = [torch.rand(B, device=dev, generator=g).requires_grad_(True) for _ in range(25)]
mode = os.environ.get("DASP_RV_NOISE", "generated")
noise = torch.randn(2 * B, 12, 65536 + 1022, device=dev, generator=g) if mode == "explicit" else None
w = torch.randn(B, 2, N, device=dev, generator=g)
def step():
    x.grad = None
    for c in ctl: c.grad = None
    if mode == "generated":
        y = D.noise_shaped_reverberation(x, 44100, *ctl, device_noise=True, noise_seed=1234)
    elif mode == "randn":
        y = D.noise_shaped_reverberation(x, 44100, *ctl, noise=torch.randn(2 * B, 12, 65536 + 1022, device=dev))
    else:
        y = D.noise_shaped_reverberation(x, 44100, *ctl, noise=noise)
    y.backward(w)
import time
for _ in range(40): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20 * 1e3
_lib.timers.start(every=1)
for _ in range(20): step()
t = _lib.timers.stop()
print(f"noise={mode} chunk={os.environ.get('DASP_REVERB_CHUNK', 'auto')} shape=({B},{C},{N}) wall {wall:.3f} ms  " + "  ".join(f"{k.replace('dasp_reverb_', '')} {sum(v) / len(v):.3f} ms" for k, v in t.items()),
      f" total {sum(sum(v) / len(v) for v in t.values()):.3f} ms  checksum {float(x.grad.double().abs().mean()):.9e} {float(ctl[0].grad.double().sum()):.6e} {float(ctl[24].grad.double().sum()):.6e}")
