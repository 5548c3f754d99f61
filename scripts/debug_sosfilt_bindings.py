"""Developer check: signal.sosfilt_via_fsm through torch.ops.dasp.sosfilt and through the ctypes binding, each against the fp64 recursion."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from oracle.recursion import sosfilt_ref, sosfilt_vjp_ref
dev = "cuda:0"
g = np.random.default_rng(3)
base = np.array([[1.0, -1.2, 0.5, 1.0, -1.5, 0.7], [0.8, 0.1, 0.2, 1.0, -0.3, 0.4], [1.1, 0.0, -0.2, 2.0, 0.4, 0.1], [0.9, 0.3, 0.1, 1.0, 0.2, 0.3],
                 [1.0, -0.5, 0.2, 1.0, -0.9, 0.5], [0.7, 0.2, -0.1, 1.0, 0.5, 0.2], [1.2, 0.1, 0.0, 1.0, -0.2, 0.6], [1.0, 0.4, 0.3, 1.0, 0.1, 0.1]])
for (B, C, N, S) in ((3, 2, 20000, 3), (3, 2, 20000, 4), (3, 2, 20000, 6), (3, 2, 20000, 2), (3, 2, 20000, 8), (3, 2, 20480, 4), (2, 1, 40000, 4), (3, 2, 5000, 3), (70, 2, 20000, 3)):
    sos = np.stack([base[:S]] * B).astype(np.float32)
    x = (g.random((B, C, N)) * 2 - 1).astype(np.float32); w = g.standard_normal((B, C, N)).astype(np.float32)
    yo = sosfilt_ref(sos.astype(np.float64), x); gxo = sosfilt_vjp_ref(sos.astype(np.float64), w)
    for flag, need_s in (("0", True), ("0", False)):
        os.environ["DASP_TORCH_OPS"] = flag
        xt = torch.from_numpy(x).to(dev).requires_grad_(True); st = torch.from_numpy(sos).to(dev).requires_grad_(need_s)
        y = D.signal.sosfilt_via_fsm(st, xt)
        (y * torch.from_numpy(w).to(dev)).sum().backward()
        rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
        e = np.abs(xt.grad.cpu().numpy() - gxo).reshape(B * C, -1)
        bad = np.argwhere(~(e < 1e-3 * np.abs(gxo).max()))
        print((B, C, N, S), "with gsos" if need_s else "gx only ", "y", "%.2e" % rel(y.detach().cpu().numpy(), yo), "gx", "%.2e" % rel(xt.grad.cpu().numpy(), gxo),
              "bad samples", len(bad), ("first (row, n) %s last %s" % (bad[0].tolist(), bad[-1].tolist())) if len(bad) else "", flush=True)
