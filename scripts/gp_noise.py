"""Developer script: distribution of parametric_eq errors vs the fp64 oracle over random EQ settings."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from oracle import dasp_oracle as orc
from tests.util import linf_peak
from tests.test_gpu_sosfilt import random_params
SR = 44100
B, C, N = 24, 2, 131072
g = np.random.default_rng(7)
x = (g.random((B, C, N), dtype=np.float32) * 2 - 1); w = g.standard_normal((B, C, N), dtype=np.float32)
p = random_params(B, 8)
p[0, 1], p[0, 2], p[0, 0] = 20.0, 6.0, 20.0
dev = "cuda:0"
xt = torch.from_numpy(x).to(dev).requires_grad_(True)
cols = [torch.from_numpy(np.ascontiguousarray(p[:, i])).to(dev).requires_grad_(True) for i in range(18)]
y = D.parametric_eq(xt, SR, *cols); (y * torch.from_numpy(w).to(dev)).sum().backward(); torch.cuda.synchronize()
gp = torch.stack([c.grad for c in cols], 1).cpu().numpy()
yo = orc.parametric_eq(x, SR, p); gxo, gpo = orc.parametric_eq_vjp(x, SR, p, w)
np.set_printoptions(linewidth=200, precision=2)
print("y ", np.sort(linf_peak(y.detach().cpu().numpy(), yo))[::-1][:8])
print("gx", np.sort(linf_peak(xt.grad.cpu().numpy(), gxo))[::-1][:8])
e = linf_peak(gp, gpo); print("gp", np.sort(e)[::-1][:8], "argmax", e.argmax())
i = e.argmax(); print("item params", p[i]); print("gp ", gp[i]); print("gpo", gpo[i])
