"""Developer timing of the multi-resolution STFT loss fwd + bwd at (16, 2, 131072) (run under rocprofv3 --kernel-trace --stats for the split)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
B, C, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 2, 131072)))
g = torch.Generator(device="cuda:0").manual_seed(0)
x = (torch.rand(B, C, N, device="cuda:0", generator=g) * 0.6 - 0.3).requires_grad_(True)
y = torch.rand(B, C, N, device="cuda:0", generator=g) * 0.6 - 0.3
fn = D.losses.MultiResolutionSTFTLoss()
def step():
    x.grad = None
    fn(x, y).backward()
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("mrstft (%d,%d,%d) fwd+bwd wall %.3f ms  loss %.6f  |g| %.6e" % (B, C, N, (time.perf_counter() - t0) / 50 * 1e3, float(fn(x, y)), float(x.grad.abs().mean())))
