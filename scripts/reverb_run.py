"""Runs noise_shaped_reverberation fwd+bwd a few times at the bench shape (profiling target)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
B, C, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 2, 262144)))
it = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = "cuda"
x = torch.rand(B, C, N, device=dev).mul_(2).sub_(1).requires_grad_(True)
ctl = [torch.rand(B, device=dev).requires_grad_(True) for _ in range(25)]
for _ in range(it):
    y = D.noise_shaped_reverberation(x, 44100, *ctl, device_noise=True)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
