"""Developer probe: control gradients of a captured compressor step after replays vs eager."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import dasp_pytorch_amd as D
B, C, N = 4, 2, int(os.environ.get("N", 65536))
g = torch.Generator(device="cuda:0").manual_seed(5)
RANGES = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]
cols = [(torch.rand(B, device="cuda:0", generator=g) * (hi - lo) + lo).requires_grad_(True) for lo, hi in RANGES]
xs = (torch.rand(B, C, N, device="cuda:0", generator=g) * 2 - 1).requires_grad_(True)
ws = torch.randn(B, C, N, device="cuda:0", generator=g)
mode = os.environ.get("MODE", "")
if "speech" in mode:
    from tests.test_gpu_dynamics import speechlike
    rng = np.random.default_rng(11)
    xs = torch.from_numpy(speechlike(rng, B, C, N)).to("cuda:0").requires_grad_(True)
if "atk" in mode:
    cols[2] = torch.full((B,), 100.0, device="cuda:0").requires_grad_(True)
if "pre" in mode:
    pre = [torch.from_numpy(speechlike(rng, B, C, N)).to("cuda:0") for _ in range(3)]
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        D.compressor(xs, 44100, *cols).backward(ws)
torch.cuda.current_stream().wait_stream(s)
xs.grad = None
for c in cols: c.grad = None
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    ys = D.compressor(xs, 44100, *cols)
    ys.backward(ws)
for k in range(3):
    xn = torch.rand(B, C, N, device="cuda:0", generator=g) * 2 - 1
    if "speech" in mode:
        xn = pre[k] if "pre" in mode else torch.from_numpy(speechlike(rng, B, C, N)).to("cuda:0")
    if "sleep" in mode:
        import time; torch.cuda.synchronize(); time.sleep(0.2)
    with torch.no_grad(): xs.copy_(xn)
    if "wcopy" in mode:
        with torch.no_grad(): ws.copy_(torch.randn(B, C, N, device="cuda:0", generator=g))
    if os.environ.get("ZERO", "1") == "1":
        xs.grad.zero_()
        for c in cols: c.grad.zero_()
    graph.replay()
    xe = xn.clone().requires_grad_(True); ce = [c.detach().clone().requires_grad_(True) for c in cols]
    D.compressor(xe, 44100, *ce).backward(ws)
    print(k, "y equal", bool(torch.equal(ys, D.compressor(xe.detach(), 44100, *[c.detach() for c in ce]))), "gx equal", bool(torch.equal(xs.grad, xe.grad)), f"{float((xs.grad - xe.grad).abs().max()):.3g}", [f"{float((a.grad - b.grad).abs().max()):.3g}/{float(b.grad.abs().max()):.3g}" for a, b in zip(cols, ce)])
