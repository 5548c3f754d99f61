"""Launch count and GPU time of the ops at the reference's training batch sizes. usage: python scripts/small_batch3.py  (run under
rocprofv3 --kernel-trace --stats to count kernels)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import _lib
SR = 44100
PEQ = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
       (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.rand(1 << 26, device="cuda"); t_end = time.perf_counter() + 0.8
while time.perf_counter() < t_end: a.mul_(1.0001)
torch.cuda.synchronize()
def gpu_ms(step, n=20):
    for _ in range(10): step()
    torch.cuda.synchronize()
    _lib.timers.start(every=1)
    for _ in range(n): step()
    kt = _lib.timers.stop()
    return sum(sum(v) for v in kt.values()) / n, {k: round(sum(v) / n, 4) for k, v in kt.items()}
for B in (8, 16, 32):
    x = (torch.rand(B, 2, 131072, device="cuda", generator=g) * 2 - 1).requires_grad_(True)
    cols = [(torch.rand(B, device="cuda", generator=g) * (hi - lo) + lo).requires_grad_(True) for lo, hi in PEQ]
    w = torch.randn(B, 2, 131072, device="cuda", generator=g)
    def step():
        x.grad = None
        for c in cols: c.grad = None
        D.parametric_eq(x, SR, *cols).backward(w)
    t, calls = gpu_ms(step)
    print(f"parametric_eq ({B},2,131072) fwd+bwd gpu {t:.4f} ms  {calls}")
chain = D.chain.StyleTransferChain(SR, device_noise=True)
xc = torch.rand(16, 1, 131072, device="cuda", generator=g) * 2 - 1
pcs = [(torch.rand(16, n, device="cuda", generator=g) * 0.9 + 0.05).requires_grad_(True) for n in chain.num_params]
wc = torch.randn(16, 2, 131072, device="cuda", generator=g)
def chain_step():
    for p in pcs: p.grad = None
    chain.process_normalized(xc, *pcs).backward(wc)
t, calls = gpu_ms(chain_step)
print(f"style_transfer_chain (16,1,131072) fwd+bwd gpu {t:.4f} ms  {calls}")
def chain_fwd():
    with torch.no_grad():
        chain.process_normalized(xc, *pcs)
t, calls = gpu_ms(chain_fwd)
print(f"style_transfer_chain (16,1,131072) forward only (no grad, fused EQ+compressor) gpu {t:.4f} ms  {calls}")
