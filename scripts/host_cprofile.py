"""Developer tool: where the host time of an eager forward + backward goes (cProfile, top entries by cumulative time), at the reference's
batch sizes. usage: python scripts/host_cprofile.py [eq|reverb|comp|chain]"""
import cProfile, os, pstats, sys, time, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
SR = 44100
what = sys.argv[1] if len(sys.argv) > 1 else "eq"
g = torch.Generator(device="cuda").manual_seed(0)
PEQ = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
       (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
B, N = 16, 131072
x = (torch.rand(B, 2, N, device="cuda", generator=g) * 2 - 1).requires_grad_(True)
w = torch.randn(B, 2, N, device="cuda", generator=g)
if what == "eq":
    cols = [(torch.rand(B, device="cuda", generator=g) * (hi - lo) + lo).requires_grad_(True) for lo, hi in PEQ]
    fn = lambda: D.parametric_eq(x, SR, *cols)
elif what == "reverb":
    cols = [torch.rand(B, device="cuda", generator=g).requires_grad_(True) for _ in range(25)]
    fn = lambda: D.noise_shaped_reverberation(x, SR, *cols, device_noise=True)
elif what == "comp":
    rng = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]
    cols = [(torch.rand(B, device="cuda", generator=g) * (hi - lo) + lo).requires_grad_(True) for lo, hi in rng]
    fn = lambda: D.compressor(x, SR, *cols)
else:
    chain = D.chain.StyleTransferChain(SR, device_noise=True)
    cols = [(torch.rand(B, n, device="cuda", generator=g) * 0.9 + 0.05).requires_grad_(True) for n in chain.num_params]
    xm = x.detach()[:, :1].contiguous()
    fn = lambda: chain.process_normalized(xm, *cols)
def step():
    x.grad = None
    for c in cols: c.grad = None
    fn().backward(w)
for _ in range(50): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print(f"{what}: wall {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:6000])
