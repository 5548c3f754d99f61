"""Developer tool: random stable filters of 4 .. 16 coefficients through lfilter_via_fsm (csrc/lfilter.hip) with random chunk lengths,
against scipy's float64 recurrence. usage: python scripts/fuzz_lfilter.py [seed]; FUZZ_SECONDS, FUZZ_DUMP=<npz of the first failure>"""
import os, sys, time
import numpy as np, torch, scipy.signal
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
t0 = time.time(); n = 0; worst = {}; dumped = False; skipped = 0
while time.time() - t0 < float(os.environ.get("FUZZ_SECONDS", "60")):
    n += 1
    B = int(rng.integers(1, 6)); N = int(rng.choice([1, 5, 16, 17, 100, 1000, 1024, 1025, 4096, 8191, 20000, int(rng.integers(1, 30000))]))
    K = int(rng.integers(4, 17)); M = K - 1
    poles = []
    while len(poles) < M:
        if M - len(poles) >= 2 and rng.random() < 0.7:
            z = rng.uniform(0.2, 0.98) * np.exp(1j * rng.uniform(0.05, 3.1)); poles += [z, np.conj(z)]
        else:
            poles.append(rng.uniform(-0.95, 0.95))
    a_ = np.real(np.poly(poles)) * rng.uniform(0.5, 2.0)
    b_ = rng.standard_normal(K) * 0.2
    bs = int(rng.choice([1, B]))
    bl = (np.tile(b_, (bs, 1)) * (1 + 0.01 * rng.standard_normal((bs, 1)))).astype(np.float32)
    al = np.tile(a_, (bs, 1)).astype(np.float32)
    rad = max(np.abs(np.roots(al[0].astype(np.float64))))            # the poles of the filter as rounded to float32: high orders are
    if rad > 0.995:                                                   # ill-conditioned in their coefficients, rounding can push a pole outside
        skipped += 1; continue
    x = (rng.random((B, 1, N)) * 2 - 1).astype(np.float32); w = rng.standard_normal((B, 1, N)).astype(np.float32)
    chunk0 = int(rng.choice([1, 5, 16, 100, 1024]))
    res = []
    for chunk in (chunk0, 10 ** 9):
        os.environ["DASP_LFILTER_CHUNK"] = str(chunk)
        xt = T(x).requires_grad_(True); bt = T(bl).requires_grad_(True); at = T(al).requires_grad_(True)
        y = D.signal.lfilter_via_fsm(xt, bt, at); (y * T(w)).sum().backward()
        res.append((y.detach().cpu().numpy(), xt.grad.cpu().numpy(), bt.grad.cpu().numpy(), at.grad.cpu().numpy()))
    yo = np.stack([scipy.signal.lfilter(bl[q % bs].astype(np.float64), al[q % bs].astype(np.float64), x[q, 0].astype(np.float64)) for q in range(B)])[:, None]
    e = dict(y=rel(res[0][0], yo), y1=rel(res[1][0], yo), gb=rel(res[0][2], res[1][2]), ga=rel(res[0][3], res[1][3]))
    for k, v in e.items(): worst[k] = max(worst.get(k, 0.0), v) if np.isfinite(v) else float("inf")
    if not (e["y"] <= 1e-5 and e["gb"] <= 1e-6 and e["ga"] <= 1e-6):
        print("FAIL", dict(B=B, N=N, K=K, bs=bs, chunk=chunk0, pole_radius=float(rad)), e, flush=True)
        if os.environ.get("FUZZ_DUMP") and not dumped:
            np.savez(os.environ["FUZZ_DUMP"], bl=bl, al=al, x=x, w=w, chunk=chunk0, y=res[0][0], y1=res[1][0], yo=yo); dumped = True
print("configs", n, "skipped (unstable after rounding)", skipped, {k: "%.2e" % v for k, v in worst.items()})
