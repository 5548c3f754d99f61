import os, sys, time, numpy as np, torch
sys.path.insert(0, ".")
import dasp_pytorch_amd as D
from oracle.recursion import sosfilt_ref
rng = np.random.default_rng(5)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def df2t32(sos, x):           # float32 transposed direct form II, section by section (what a plain fp32 implementation does)
    y = x.astype(np.float32).copy()
    B, C, N = y.shape
    for k in range(sos.shape[1]):
        c = (sos[:, k] / sos[:, k, 3:4]).astype(np.float32)
        b0, b1, b2, a1, a2 = (c[:, i][:, None] for i in (0, 1, 2, 4, 5))
        z1 = np.zeros((B, C), np.float32); z2 = np.zeros((B, C), np.float32)
        for n in range(N):
            u = y[:, :, n]
            o = b0 * u + z1
            z1 = b1 * u - a1 * o + z2
            z2 = b2 * u - a2 * o
            y[:, :, n] = o
    return y
for Sx, Bq in ((12, 70), (7, 70), (8, 70), (12, 3), (6, 70)):
    for rep in range(3):
        Cq, Nq = 2, 3000
        rr = 0.2 + 0.75 * rng.random((Bq, Sx)); th = 3.0 * rng.random((Bq, Sx)) + 0.05
        sq = np.zeros((Bq, Sx, 6)); sq[..., :3] = rng.standard_normal((Bq, Sx, 3)) * 0.7
        sq[..., 3] = 1.0 + 0.2 * rng.random((Bq, Sx)); sq[..., 4] = -2 * rr * np.cos(th) * sq[..., 3]; sq[..., 5] = rr * rr * sq[..., 3]
        sq = sq.astype(np.float32)
        xq = (rng.random((Bq, Cq, Nq)) * 2 - 1).astype(np.float32)
        with torch.no_grad():
            y = D.signal.sosfilt_via_fsm(T(sq), T(xq)).cpu().numpy()
        sqn = sq.astype(np.float64) / sq[..., 3:4].astype(np.float64)
        yo = sosfilt_ref(sqn, xq)
        y32 = df2t32(sq, xq)
        pk = np.abs(yo).max(-1)
        ek = (np.abs(y - yo).max(-1) / pk); e32 = (np.abs(y32 - yo).max(-1) / pk)
        i = np.unravel_index(ek.argmax(), ek.shape)
        print(f"S {Sx} rows {Bq * Cq}: kernels worst row {ek.max():.1e} (plain fp32 on that row {e32[i]:.1e}); plain fp32 worst row {e32.max():.1e}; median rows {np.median(ek):.1e} / {np.median(e32):.1e}", flush=True)
