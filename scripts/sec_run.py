"""Compressor (256,2,262144) and reverb (128,2,262144) fwd+bwd, a few iterations: the target of the FETCH_SIZE / WRITE_SIZE passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(256, 2, 262144, device=dev, generator=g) * 2 - 1).requires_grad_(True)
rng = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]
ctl = [(torch.rand(256, device=dev, generator=g) * (hi - lo) + lo).requires_grad_(True) for lo, hi in rng]
xr = (torch.rand(128, 2, 262144, device=dev, generator=g) * 2 - 1).requires_grad_(True)
cr = [torch.rand(128, device=dev, generator=g).requires_grad_(True) for _ in range(25)]
for _ in range(3):
    y = D.compressor(x, 44100, *ctl); y.backward(torch.ones_like(y))
    y = D.noise_shaped_reverberation(xr, 44100, *cr, device_noise=True); y.backward(torch.ones_like(y))
torch.cuda.synchronize()
