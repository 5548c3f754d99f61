"""Developer tool: randomized shape sweep of every op against the numpy oracle (forward and all gradients) on small problems.
Catches layout / tail / alignment slips in the tile paths that the fixed-shape tests might miss."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from oracle import dasp_oracle as orc
from oracle.recursion import sosfilt_ref, sosfilt_vjp_ref
SR = 44100
dev = "cuda:0"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
PEQ = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
       (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
lo = np.array([r[0] for r in PEQ]); hi = np.array([r[1] for r in PEQ])
worst = {}
eq_rows = []
def rel(a, b, floor=1e-30):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor))
def note(op, key, v, tol, cfg):
    worst[(op, key)] = max(worst.get((op, key), 0.0), v)
    if not (v <= tol):
        print("FAIL", op, key, v, "tol", tol, cfg, flush=True)
def randN():
    return int(rng.choice([1, 3, 4, 63, 64, 1000, 1023, 1024, 1025, 2047, 4096, 4100, 8191, 8192, 9000, 16385, 20000, int(rng.integers(1, 40000))]))
t0 = time.time()
n_cfg = 0
while time.time() - t0 < float(os.environ.get("FUZZ_SECONDS", "120")):
    n_cfg += 1
    B, C, N = int(rng.integers(1, 5)), int(rng.integers(1, 4)), randN()
    x = (rng.random((B, C, N)) * 2 - 1).astype(np.float32); w = rng.standard_normal((B, C, N)).astype(np.float32)
    cfg = (B, C, N)
    # parametric_eq (N >= 8192 against the oracle of the reference; shorter: against the exact recursion through signal.biquad)
    p = (rng.random((B, 18)) * (hi - lo) + lo).astype(np.float32)
    need_x = bool(rng.random() < 0.7)                   # the no-gx backward variant when x needs no gradient
    xt = T(x).requires_grad_(need_x); cols = [T(p[:, i]).requires_grad_(True) for i in range(18)]
    y = D.parametric_eq(xt, SR, *cols); (y * T(w)).sum().backward()
    sos = orc.peq_sos(p.astype(np.float64), SR)
    yo = sosfilt_ref(sos, x)
    gxo = sosfilt_vjp_ref(sos, w)
    note("eq", "y", rel(y.detach().cpu().numpy(), yo), 2e-5, cfg)
    if need_x:
        note("eq", "gx", rel(xt.grad.cpu().numpy(), gxo), 3e-5, cfg)
    if N >= 256:
        # control gradients against the float64 path (csrc/ref64.hip: the same recursion in double; pinned to the reference's fp64 goldens
        # and to gradcheck by tests/test_gpu_fp64.py) - valid at any length, unlike the reference's circular method, which aliases the
        # impulse-response tail of short signals
        c64 = [T(p[:, i].astype(np.float64)).requires_grad_(True) for i in range(18)]
        (D.parametric_eq(T(x.astype(np.float64)), SR, *c64) * T(w.astype(np.float64))).sum().backward()
        gpo = torch.stack([c.grad for c in c64], 1).cpu().numpy()
        gp = torch.stack([c.grad for c in cols], 1).cpu().numpy()
        erow = np.abs(gp - gpo).max(1) / np.abs(gpo).max(1)
        eq_rows.extend(erow.tolist())
        if erow.max() > worst.get(("eq", "gparams"), 0.0):
            k = int(erow.argmax()); col = int(np.abs(gp - gpo)[k].argmax())
            worst_eq = dict(cfg=cfg, need_x=need_x, err=float(erow.max()), column=col, params=[round(float(v), 4) for v in p[k]],
                            g64=[float("%.3e" % v) for v in gpo[k]], g32=[float("%.3e" % v) for v in gp[k]])
        note("eq", "gparams", float(erow.max()), 2e-4, cfg + (need_x,))
    if os.environ.get("FUZZ_EQ_ONLY"):
        continue
    # gain / distortion
    for name, fn, f, fv, shape in (("gain", D.gain, orc.gain, orc.gain_vjp, (B,)), ("dist", D.distortion, orc.distortion, orc.distortion_vjp, (B * C,))):
        c = (rng.random(shape) * 24).astype(np.float32)
        xt = T(x).requires_grad_(True); ct = T(c).requires_grad_(True)
        y = fn(xt, SR, ct); (y * T(w)).sum().backward()
        gxo, gco = fv(x, SR, c, w)
        # floor for the control gradient: a signed sum of N terms that nearly cancels is ill-conditioned relative to itself; its error is
        # measured against the size such a sum typically has (sqrt(terms) * |w|), not against a near-zero result
        fl = 0.05 * float(np.abs(w).max()) * float(np.sqrt(x.size / c.size))
        note(name, "y", rel(y.detach().cpu().numpy(), f(x, SR, c)), 2e-6, cfg); note(name, "gx", rel(xt.grad.cpu().numpy(), gxo), 3e-6 if name == "gain" else 2e-5, cfg)       # sech^2(u) moves by 2 |u| eps with the rounding of u
        note(name, "gc", rel(ct.grad.cpu().numpy(), gco, fl), 1e-4, cfg)
    # compressor (signals bounded away from silence so that the gain computer's kinks are not sampled exactly)
    if N >= 8192 or True:
        pc = np.stack([rng.random(B) * 60 - 60, rng.random(B) * 19 + 1, rng.random(B) * 95 + 5, rng.random(B) * 95 + 5, rng.random(B) * 12 + 1e-3, rng.random(B) * 12], 1).astype(np.float32)
        look = int(rng.choice([0, 0, 0, 5]))
        xt = T(x).requires_grad_(True); cc = [T(pc[:, i]).requires_grad_(True) for i in range(6)]
        y = D.compressor(xt, SR, *cc, lookahead_samples=look); (y * T(w)).sum().backward()
        pd = pc.astype(np.float64)
        from oracle.recursion import one_pole_ref
        c = orc._compressor_core(x, SR, pd[:, 0], pd[:, 1], pd[:, 2], pd[:, 4], pd[:, 5], 1e-8, look, np.float64)
        g = one_pole_ref(c["g_c"][:, 0], c["alpha"][:, 0, 0])[:, None]
        yo = c["x_d"] * 10 ** ((g + c["mk"]) / 20)
        note("comp", "y", rel(y.detach().cpu().numpy(), yo), 3e-5, cfg + (look,))
        assert torch.isfinite(xt.grad).all() and all(torch.isfinite(q.grad).all() for q in cc)
    # fused EQ -> compressor forward (csrc/chainfwd.hip) on mono / stereo x, plain and with forced segment lengths, against the exact
    # recursions (EQ through the fp64 sos recursion, compressor core + one-pole recursion): valid at any length
    if C <= 2:
        from dasp_pytorch_amd import ops as _ops
        from dasp_pytorch_amd.functional import _PEQ_TYPES
        pn = rng.random((B, 18)).astype(np.float32)
        pq = (pn.astype(np.float64) * (hi - lo) + lo)
        tiles = rng.choice([0, 0, 16, 32])
        D.config.plan.chain_segment_tiles = int(tiles) if tiles else None
        ctl = np.stack([pc[:, 0], pc[:, 1], pc[:, 2], pc[:, 4], pc[:, 5]], 1).astype(np.float32)
        with torch.no_grad():
            yf = _ops.chain_eq_compressor_forward(T(x), T(pn), _PEQ_TYPES, [float(v) for v in lo], [float(v) for v in hi - lo], float(SR), T(ctl))
        D.config.plan.chain_segment_tiles = None
        y1 = sosfilt_ref(orc.peq_sos(pq, SR), x)
        c2 = orc._compressor_core(y1, SR, pd[:, 0], pd[:, 1], pd[:, 2], pd[:, 4], pd[:, 5], 1e-8, 0, np.float64)
        g2 = one_pole_ref(c2["g_c"][:, 0], c2["alpha"][:, 0, 0])[:, None]
        note("chainfwd", "y", rel(yf.cpu().numpy(), c2["x_d"] * 10 ** ((g2 + c2["mk"]) / 20)), 5e-5, cfg + (int(tiles),))
    # stereo utilities
    Tn = int(rng.integers(1, 6))
    xw = (rng.random((B, 2, N)) * 2 - 1).astype(np.float32); wd = rng.random((B, 1)).astype(np.float32)
    xp = (rng.random((B, Tn, N)) * 2 - 1).astype(np.float32); pan = (rng.random((B, Tn)) * 0.9 + 0.05).astype(np.float32)
    xb = (rng.random((B, 2, Tn, N)) * 2 - 1).astype(np.float32); send = (rng.random((B, Tn, 1)) * 36 - 24).astype(np.float32)
    for name, fn, f, fv, xx, c, osh in (("wid", D.stereo_widener, orc.stereo_widener, orc.stereo_widener_vjp, xw, wd, (B, 2, N)),
                                       ("pan", D.stereo_panner, orc.stereo_panner, orc.stereo_panner_vjp, xp, pan, (B, 2, Tn, N)),
                                       ("bus", D.stereo_bus, orc.stereo_bus, orc.stereo_bus_vjp, xb, send, (B, 2, N))):
        ww = rng.standard_normal(osh).astype(np.float32)
        xt = T(xx).requires_grad_(True); ct = T(c).requires_grad_(True)
        y = fn(xt, SR, ct); (y * T(ww)).sum().backward()
        gxo, gco = fv(xx, SR, c, ww)
        # (gx of the panner / widener is a two-term signed sum per sample: on a one-sample signal it can cancel, hence the floor)
        note(name, "y", rel(y.detach().cpu().numpy(), f(xx, SR, c)), 3e-6, cfg + (Tn,)); note(name, "gx", rel(xt.grad.cpu().numpy(), gxo, 0.05 * float(np.abs(ww).max())), 3e-6, cfg + (Tn,))
        note(name, "gc", rel(ct.grad.cpu().numpy(), gco, 0.05 * float(np.abs(ww).max()) * float(np.sqrt(xx.size / c.size))), 2e-4, cfg + (Tn,))
    # sosfilt_via_fsm with 1 .. 8 sections (one call per direction) on few and many rows (from ~128 rows on a row has a workgroup of its own: for 7 / 8 sections
    # that is the checkpointed backward kernel, for fewer rows the segmented kernels): y and grad x against the fp64 recursion, the
    # coefficient gradients against the float64 path. Random cascades are far worse conditioned than the EQ's (a resonance in one section,
    # a zero next to it in another: intermediate signals dwarf the output) - the bounds are those a plain fp32 recursion meets on the
    # same inputs (scripts/sections_accuracy.py)
    if n_cfg % 4 == 0:
        Sx = int(rng.integers(1, 9)); Bq = int(rng.choice([1, 3, 40, 70, 100])); Cq = int(rng.integers(1, 3)); Nq = min(randN(), 6000)
        rr = 0.2 + 0.75 * rng.random((Bq, Sx)); th = 3.0 * rng.random((Bq, Sx)) + 0.05
        sq = np.zeros((Bq, Sx, 6)); sq[..., :3] = rng.standard_normal((Bq, Sx, 3)) * 0.7
        sq[..., 3] = 1.0 + 0.2 * rng.random((Bq, Sx)); sq[..., 4] = -2 * rr * np.cos(th) * sq[..., 3]; sq[..., 5] = rr * rr * sq[..., 3]
        sq = sq.astype(np.float32)
        xq = (rng.random((Bq, Cq, Nq)) * 2 - 1).astype(np.float32); wq = rng.standard_normal((Bq, Cq, Nq)).astype(np.float32)
        needx = bool(rng.random() < 0.7)
        xt = T(xq).requires_grad_(needx); st = T(sq).requires_grad_(True)
        y = D.signal.sosfilt_via_fsm(st, xt); (y * T(wq)).sum().backward()
        sqn = sq.astype(np.float64) / sq[..., 3:4].astype(np.float64)
        note("sos", "y", rel(y.detach().cpu().numpy(), sosfilt_ref(sqn, xq)), 3e-4, (Bq, Cq, Nq, Sx))
        if needx:
            note("sos", "gx", rel(xt.grad.cpu().numpy(), sosfilt_vjp_ref(sqn, wq)), 3e-4, (Bq, Cq, Nq, Sx))
        s64 = T(sq.astype(np.float64)).requires_grad_(True)
        (D.signal.sosfilt_via_fsm(s64, T(xq.astype(np.float64))) * T(wq.astype(np.float64))).sum().backward()
        gref = s64.grad.cpu().numpy()
        note("sos", "gsos", float(np.abs(st.grad.cpu().numpy() - gref).max() / np.abs(gref).max()), 1e-3, (Bq, Cq, Nq, Sx, needx))
    # reverb (small impulse responses, odd shapes)
    if n_cfg % 3 == 0:
        L = int(rng.choice([64, 300, 1000, 2048, 3000, 5000, 9000])); taps = int(rng.choice([15, 63, 127, 1023]))
        Cr = int(rng.integers(1, 3)); xr = (rng.random((B, Cr, N)) * 2 - 1).astype(np.float32); wr = rng.standard_normal((B, 2, N)).astype(np.float32)
        pr = rng.random((B, 25)).astype(np.float32); noise = rng.standard_normal((2 * B, 12, L + taps - 1)).astype(np.float32)
        xt = T(xr).requires_grad_(True); cr = [T(pr[:, i]).requires_grad_(True) for i in range(25)]
        y = D.noise_shaped_reverberation(xt, SR, *cr, num_samples=L, num_bandpass_taps=taps, noise=T(noise)); (y * T(wr)).sum().backward()
        pd = pr.astype(np.float64)
        yo = orc.noise_shaped_reverberation(xr, SR, pd[:, :12], pd[:, 12:24], pd[:, 24], noise, L, taps)
        gxo, gg, gd, gm = orc.noise_shaped_reverberation_vjp(xr, SR, pd[:, :12], pd[:, 12:24], pd[:, 24], noise, wr, L, taps)
        note("rev", "y", rel(y.detach().cpu().numpy(), yo), 5e-5, cfg + (Cr, L, taps)); note("rev", "gx", rel(xt.grad.cpu().numpy(), gxo), 5e-5, cfg + (Cr, L, taps))
        gp = torch.stack([q.grad for q in cr], 1).cpu().numpy(); gpo = np.concatenate([gg, gd, gm[:, None]], 1)
        note("rev", "gp", float(np.abs(gp - gpo).max() / np.abs(gpo).max()), 5e-4, cfg + (Cr, L, taps))
        # the same call with the noise generated inside the kernels: the stream written out and fed back must reproduce it
        from dasp_pytorch_amd import ops as _ops
        seed = int(rng.integers(0, 2 ** 62))
        nz = _ops.reverb_noise(seed, B, 12, L + taps - 1, dev)
        outs = []
        for kw in (dict(noise=nz), dict(device_noise=True, noise_seed=seed)):
            xt = T(xr).requires_grad_(True); cr = [T(pr[:, i]).requires_grad_(True) for i in range(25)]
            y = D.noise_shaped_reverberation(xt, SR, *cr, num_samples=L, num_bandpass_taps=taps, **kw); (y * T(wr)).sum().backward()
            outs.append((y.detach().cpu().numpy(), xt.grad.cpu().numpy(), torch.stack([q.grad for q in cr], 1).cpu().numpy()))
        note("rev_gen", "y", rel(outs[1][0], outs[0][0]), 1e-5, cfg + (Cr, L, taps)); note("rev_gen", "gx", rel(outs[1][1], outs[0][1]), 3e-6, cfg + (Cr, L, taps))
        note("rev_gen", "gp", rel(outs[1][2], outs[0][2]), 2e-5, cfg + (Cr, L, taps))
print("configs", n_cfg)
if "worst_eq" in globals(): print("worst eq control-gradient row:", worst_eq)
if eq_rows:
    e = np.sort(np.array(eq_rows))
    print("eq control gradients vs the fp64 path, error / largest entry of the item's 18 gradients: items %d  median %.1e  p99 %.1e  p99.9 %.1e  max %.1e  above 1e-4: %.2f %%"
          % (len(e), e[len(e) // 2], e[int(0.99 * len(e))], e[int(0.999 * len(e))], e[-1], 100.0 * float((e > 1e-4).mean())))
for k in sorted(worst): print(k, "%.2e" % worst[k])
