"""Developer timing: parametric_eq forward + backward at small batches, plain rows vs segmented rows (config.plan.sos_segment False / True),
plus the difference of the two paths' results."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import config
SR = 44100
R = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
     (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
g = torch.Generator(device="cuda:0").manual_seed(0)


def run(B, C, N, force_plain):
    config.plan.sos_segment = not force_plain
    x = (torch.rand(B, C, N, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1)) * 2 - 1).requires_grad_(True)
    gg = torch.Generator(device="cuda:0").manual_seed(2)
    cols = [(torch.rand(B, device="cuda:0", generator=gg) * (hi - lo) + lo).requires_grad_(True) for lo, hi in R]
    w = torch.randn(B, C, N, device="cuda:0", generator=gg)

    def step():
        x.grad = None
        for c in cols:
            c.grad = None
        y = D.parametric_eq(x, SR, *cols)
        y.backward(w)
        return y
    for _ in range(20):
        y = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    return dt, y.detach().clone(), x.grad.clone(), torch.stack([c.grad for c in cols], 1).clone()


for B, C, N in ((4, 2, 131072), (16, 2, 131072), (16, 1, 131072), (32, 2, 131072), (8, 2, 50000)):
    tp, yp, gxp, gpp = run(B, C, N, True)
    ts, ys, gxs, gps = run(B, C, N, False)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print(f"({B},{C},{N}): plain {tp*1e3:.3f} ms  segmented {ts*1e3:.3f} ms   |dy| {rel(ys, yp):.2e}  |dgx| {rel(gxs, gxp):.2e}  |dgp| {rel(gps, gpp):.2e}", flush=True)

# GPU-side times of the entry points (HIP events around every call; the wall times above are launch-bound at these sizes)
from dasp_pytorch_amd import _lib
for force_plain in (True, False):
    config.plan.sos_segment = not force_plain
    B, C, N = 16, 2, 131072
    x = (torch.rand(B, C, N, device="cuda:0") * 2 - 1).requires_grad_(True)
    cols = [(torch.rand(B, device="cuda:0") * (hi - lo) + lo).requires_grad_(True) for lo, hi in R]
    w = torch.randn(B, C, N, device="cuda:0")
    for _ in range(30):
        D.parametric_eq(x, SR, *cols).backward(w)
    torch.cuda.synchronize()
    _lib.timers.start()
    for _ in range(50):
        D.parametric_eq(x, SR, *cols).backward(w)
    kt = _lib.timers.stop()
    print("plain" if force_plain else "segmented", {k: round(sum(v) / len(v), 4) for k, v in kt.items()}, flush=True)
