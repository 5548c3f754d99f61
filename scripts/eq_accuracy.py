"""Developer check: accuracy of parametric_eq forward / grad_x against the fp64 oracle for random controls and for the worst corner of
the ParametricEQ ranges (lowest cut-off frequencies, highest Q: poles closest to z = 1)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from oracle import dasp_oracle as orc
from tests.util import linf_peak
SR = 44100
R = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
     (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
rng = np.random.default_rng(0)
B, C, N = 8, 2, 65536
x = (rng.random((B, C, N)) * 2 - 1).astype(np.float32)
w = rng.standard_normal((B, C, N)).astype(np.float32)
lo = np.array([r[0] for r in R]); hi = np.array([r[1] for r in R])
p = (rng.random((B, 18)) * (hi - lo) + lo).astype(np.float32)
for b in range(4):            # corner items: min cut-off, max Q, gains at +-20 dB
    p[b, 1::3] = lo[1::3]; p[b, 2::3] = hi[2::3]; p[b, 0::3] = 20.0 if b % 2 else -20.0
p[2, 2::3] = lo[2::3]; p[3, 2::3] = lo[2::3]   # two of them with the lowest Q instead
for b in (4, 5):              # and the opposite corner: max cut-off (poles nearest z = -1), max / min Q
    p[b, 1::3] = hi[1::3]; p[b, 2::3] = hi[2::3] if b == 4 else lo[2::3]; p[b, 0::3] = 20.0 if b % 2 else -20.0
xt = torch.from_numpy(x).cuda().requires_grad_(True)
cols = [torch.from_numpy(p[:, i].copy()).cuda().requires_grad_(True) for i in range(18)]
y = D.parametric_eq(xt, SR, *cols)
(y * torch.from_numpy(w).cuda()).sum().backward()
yo = orc.parametric_eq(x, SR, p.astype(np.float64))
gxo, gpo = orc.parametric_eq_vjp(x, SR, p.astype(np.float64), w)
gp = torch.stack([c.grad for c in cols], 1).cpu().numpy()
print("y   err per item", np.array2string(linf_peak(y.detach().cpu().numpy(), yo), precision=2))
print("gx  err per item", np.array2string(linf_peak(xt.grad.cpu().numpy(), gxo), precision=2))
print("gp  err per item", np.array2string(linf_peak(gp, gpo), precision=2))
