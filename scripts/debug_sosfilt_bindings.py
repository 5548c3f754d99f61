"""Developer check: signal.sosfilt_via_fsm through torch.ops.dasp.sosfilt and through the ctypes binding, each against the fp64 recursion."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from oracle.recursion import sosfilt_ref, sosfilt_vjp_ref
dev = "cuda:0"
g = np.random.default_rng(3)
for (B, C, N) in ((3, 2, 20000), (3, 2, 5000), (70, 2, 20000)):
    sos = np.stack([np.array([[1.0, -1.2, 0.5, 1.0, -1.5, 0.7], [0.8, 0.1, 0.2, 1.0, -0.3, 0.4], [1.1, 0.0, -0.2, 2.0, 0.4, 0.1]])] * B).astype(np.float32)
    x = (g.random((B, C, N)) * 2 - 1).astype(np.float32); w = g.standard_normal((B, C, N)).astype(np.float32)
    yo = sosfilt_ref(sos.astype(np.float64), x); gxo = sosfilt_vjp_ref(sos.astype(np.float64), w)
    for flag in ("1", "0", "1"):
        os.environ["DASP_TORCH_OPS"] = flag
        xt = torch.from_numpy(x).to(dev).requires_grad_(True); st = torch.from_numpy(sos).to(dev).requires_grad_(True)
        y = D.signal.sosfilt_via_fsm(st, xt)
        (y * torch.from_numpy(w).to(dev)).sum().backward()
        rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
        print((B, C, N), "torch_ops" if flag == "1" else "ctypes", "y", "%.2e" % rel(y.detach().cpu().numpy(), yo), "gx", "%.2e" % rel(xt.grad.cpu().numpy(), gxo),
              "gsos[0,0]", st.grad[0, 0].cpu().numpy().round(3).tolist(), flush=True)
