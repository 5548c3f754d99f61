"""Generate golden vectors by running the *reference itself* (csteinmetz1/dasp-pytorch v0.0.1,
imported from /root/reference) on CPU. The reference ships no tests or fixtures, so these are the
pins for oracle/ and for the HIP kernels. Run in the build container only:

    python tests/golden/make_golden.py

Outputs are stored as float32 (the fp64-reference results are rounded to fp32 on save; that costs
6e-8 relative, far below every tolerance used). Seeds are fixed; re-running reproduces the files.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import dasp_pytorch  # noqa: E402
import dasp_pytorch.functional as RF  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SR = 44100


def f32(t):
    return t.detach().to(torch.float32).numpy()


def denorm(mod, p):
    return mod.denormalize_param_dict(mod.extract_param_dict(p))


def run_eq(x, params18, w, dtype):
    x = x.to(dtype).clone().requires_grad_(True)
    cols = [params18[:, i].to(dtype).clone().requires_grad_(True) for i in range(18)]
    y = RF.parametric_eq(x, SR, *cols)
    (y * w.to(dtype)).sum().backward()
    gp = torch.stack([c.grad for c in cols], 1)
    return y, x.grad, gp


def eq_case(name, B, C, N, Bp, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, C, N, generator=g) * 2 - 1
    pn = torch.rand(Bp, 18, generator=g)
    mod = dasp_pytorch.ParametricEQ(SR)
    d = denorm(mod, pn)
    params = torch.stack([d[k] for k in mod.param_ranges.keys()], 1)   # (Bp, 18) physical units
    w = torch.randn(B, C, N, generator=g)
    out = dict(x=f32(x), params=f32(params), w=f32(w))
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        y, gx, gp = run_eq(x, params, w, dt)
        out["y" + tag], out["gx" + tag], out["gp" + tag] = f32(y), f32(gx), f32(gp)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


def sos_case(name, B, C, N, S, seed):
    """Direct sosfilt_via_fsm boundary with generic (non-unit a0) stable sections."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, C, N, generator=g) * 2 - 1
    r = 0.3 + 0.69 * torch.rand(B, S, generator=g)
    th = 3.1 * torch.rand(B, S, generator=g)
    a0 = 0.5 + torch.rand(B, S, generator=g)
    a = torch.stack([torch.ones_like(r), -2 * r * torch.cos(th), r * r], -1) * a0[..., None]
    b = torch.randn(B, S, 3, generator=g)
    # make some sections have real poles
    a[:, 0, 1] = -(0.9 + 0.5) * a0[:, 0]
    a[:, 0, 2] = (0.9 * 0.5) * a0[:, 0]
    sos = torch.cat([b, a], -1)
    w = torch.randn(B, C, N, generator=g)
    out = dict(x=f32(x), sos=f32(sos), w=f32(w))
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        xx = x.to(dt).clone().requires_grad_(True)
        ss = sos.to(torch.float32).to(dt).clone().requires_grad_(True)   # both precisions consume the fp32-rounded sos
        y = dasp_pytorch.signal.sosfilt_via_fsm(ss, xx)
        (y * w.to(dt)).sum().backward()
        out["y" + tag], out["gx" + tag], out["gsos" + tag] = f32(y), f32(xx.grad), f32(ss.grad)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


def gain_dist_case(name, B, C, N, seed):
    """BASELINE config 1: distortion() + gain() on random (4,1,16384)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, C, N, generator=g) * 2 - 1
    gain_db = torch.rand(B, generator=g) * 48 - 24
    drive_db = torch.rand(B * C, generator=g) * 24
    w = torch.randn(B, C, N, generator=g)
    out = dict(x=f32(x), gain_db=f32(gain_db), drive_db=f32(drive_db), w=f32(w))
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        xx = x.to(dt).clone().requires_grad_(True)
        gd = gain_db.to(dt).clone().requires_grad_(True)
        y = RF.gain(xx, SR, gd)
        (y * w.to(dt)).sum().backward()
        out["gain_y" + tag], out["gain_gx" + tag], out["gain_gp" + tag] = f32(y), f32(xx.grad), f32(gd.grad)
        xx = x.to(dt).clone().requires_grad_(True)
        dd = drive_db.to(dt).clone().requires_grad_(True)
        y = RF.distortion(xx, SR, dd)
        (y * w.to(dt)).sum().backward()
        out["dist_y" + tag], out["dist_gx" + tag], out["dist_gp" + tag] = f32(y), f32(xx.grad), f32(dd.grad)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


def dist_sample_case(name, B, C, N, seed):
    """distortion() with one drive value per sample: drive_db (B, C, N), the other shape functional.py:78's view(bs, chs, -1) accepts."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, C, N, generator=g) * 2 - 1
    drive_db = torch.rand(B, C, N, generator=g) * 24
    w = torch.randn(B, C, N, generator=g)
    out = dict(x=f32(x), drive_db=f32(drive_db), w=f32(w))
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        xx = x.to(dt).clone().requires_grad_(True)
        dd = drive_db.to(dt).clone().requires_grad_(True)
        y = RF.distortion(xx, SR, dd)
        (y * w.to(dt)).sum().backward()
        out["y" + tag], out["gx" + tag], out["gp" + tag] = f32(y), f32(xx.grad), f32(dd.grad)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


def stereo_case(name, B, T, N, seed):
    """stereo_widener (B,2,N), stereo_panner (B,T,N) -> (B,2,T,N), stereo_bus (B,2,T,N) -> (B,2,N): forward and all gradients."""
    g = torch.Generator().manual_seed(seed)
    xw = torch.rand(B, 2, N, generator=g) * 2 - 1
    width = torch.rand(B, 1, generator=g)                      # the reference's broadcast needs (bs, 1), its docstring says (bs)
    xp = torch.rand(B, T, N, generator=g) * 2 - 1
    pan = torch.rand(B, T, generator=g) * 0.9 + 0.05          # away from 0 / 1, where the reference's sqrt has an infinite slope
    xb = torch.rand(B, 2, T, N, generator=g) * 2 - 1
    send = torch.rand(B, T, 1, generator=g) * 36 - 24
    ww, wp, wb = torch.randn(B, 2, N, generator=g), torch.randn(B, 2, T, N, generator=g), torch.randn(B, 2, N, generator=g)
    out = dict(xw=f32(xw), width=f32(width), xp=f32(xp), pan=f32(pan), xb=f32(xb), send=f32(send), ww=f32(ww), wp=f32(wp), wb=f32(wb))
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        for key, fn, x, c, w in (("wid", RF.stereo_widener, xw, width, ww), ("pan", RF.stereo_panner, xp, pan, wp),
                                 ("bus", RF.stereo_bus, xb, send, wb)):
            xx = x.to(dt).clone().requires_grad_(True)
            cc = c.to(dt).clone().requires_grad_(True)
            y = fn(xx, SR, cc)
            (y * w.to(dt)).sum().backward()
            out[key + "_y" + tag], out[key + "_gx" + tag], out[key + "_gc" + tag] = f32(y), f32(xx.grad), f32(cc.grad)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


COMP_KEYS = ["threshold_db", "ratio", "attack_ms", "release_ms", "knee_db", "makeup_gain_db"]


def comp_case(name, B, C, N, seed, lookahead=0, speechlike=False):
    """compressor fwd + all gradients (BASELINE config 3 at a size the CPU reference runs in seconds).
    knee is kept >= 1e-3 (the reference's backward is NaN at knee_db == 0)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, C, N, generator=g) * 2 - 1
    if speechlike:   # slow random envelope spanning -60..0 dBFS so that all three knee regions are exercised
        env_db = torch.nn.functional.interpolate(torch.rand(B, 1, N // 500 + 2, generator=g) * 60 - 60, size=N, mode="linear")
        x = x * 10 ** (env_db / 20)
    pn = torch.rand(B, 6, generator=g)
    pn[:, 4] = pn[:, 4].clamp_min(1e-3 / 12)
    mod = dasp_pytorch.Compressor(SR)
    d = denorm(mod, pn)
    params = torch.stack([d[k] for k in COMP_KEYS], 1)
    w = torch.randn(B, C, N, generator=g)
    out = dict(x=f32(x), params=f32(params), w=f32(w), lookahead=np.int64(lookahead))
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        xx = x.to(dt).clone().requires_grad_(True)
        cols = [params[:, i].to(dt).clone().requires_grad_(True) for i in range(6)]
        y = RF.compressor(xx, SR, *cols, lookahead_samples=lookahead)
        (y * w.to(dt)).sum().backward()
        gp = torch.stack([c.grad if c.grad is not None else torch.zeros_like(c) for c in cols], 1)
        out["y" + tag], out["gx" + tag], out["gp" + tag] = f32(y), f32(xx.grad), f32(gp)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


def reverb_case(name, B, C, N, L, taps, seed, store_noise):
    """noise_shaped_reverberation fwd + all gradients. The reference draws its noise from the global CPU generator
    (functional.py:548): the golden records the seed set immediately before each call, plus either the noise itself
    (small cases) or a checksum of it."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, C, N, generator=g) * 2 - 1
    p = torch.rand(B, 25, generator=g)           # gains 12, decays 12, mix  (modules.py:204-230 ranges are all [0, 1])
    w = torch.randn(B, 2, N, generator=g)
    nseed = 4321 + seed
    torch.manual_seed(nseed)
    noise = torch.randn(B * 2, 12, L + taps - 1)
    out = dict(x=f32(x), params=f32(p), w=f32(w), L=np.int64(L), taps=np.int64(taps), noise_seed=np.int64(nseed),
               noise_sum=np.float64(noise.double().sum().item()), noise_head=f32(noise[0, 0, :8]))
    if store_noise:
        out["noise"] = f32(noise)
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        xx = x.to(dt).clone().requires_grad_(True)
        cols = [p[:, i].to(dt).clone().requires_grad_(True) for i in range(25)]
        torch.manual_seed(nseed)
        y = RF.noise_shaped_reverberation(xx, SR, *cols, num_samples=L, num_bandpass_taps=taps)
        (y * w.to(dt)).sum().backward()
        out["y" + tag], out["gx" + tag] = f32(y), f32(xx.grad)
        out["gp" + tag] = f32(torch.stack([c.grad for c in cols], 1))
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items()})


def norm_case(name, modname, B, C, N, seed, noise_seed=None):
    """Processor.process_normalized (dasp_pytorch/modules.py:25-51, 70-91) on a normalised (B, P) parameter tensor: forward, grad x and
    the gradient w.r.t. the normalised parameters themselves - the pin for dasp_pytorch_amd.modules."""
    g = torch.Generator().manual_seed(seed)
    mod = getattr(dasp_pytorch, modname)(SR)
    x = torch.rand(B, C, N, generator=g) * 2 - 1
    pn = torch.rand(B, mod.num_params, generator=g)
    if modname == "Compressor":
        pn[:, 4] = pn[:, 4].clamp_min(1e-3 / 12)          # knee_db > 0: the reference's backward is NaN at 0
    oc = 2 if modname == "NoiseShapedReverb" else C
    w = torch.randn(B, oc, N, generator=g)
    out = dict(x=f32(x), pn=f32(pn), w=f32(w))
    if noise_seed is not None:
        out["noise_seed"] = np.int64(noise_seed)
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        xx = x.to(dt).clone().requires_grad_(True)
        pp = pn.to(dt).clone().requires_grad_(True)
        if noise_seed is not None:
            torch.manual_seed(noise_seed)
        y = mod.process_normalized(xx, pp)
        (y * w.to(dt)).sum().backward()
        out["y" + tag], out["gx" + tag], out["gpn" + tag] = f32(y), f32(xx.grad), f32(pp.grad)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items()})


BIQUAD_TYPES = ["peaking", "low_shelf", "high_shelf", "low_pass", "high_pass"]


def biquad_case(name, B, seed):
    """signal.biquad (dasp_pytorch/signal.py:242-306), all five filter types: coefficients and the gradients of a random linear
    functional of them w.r.t. gain_db, cutoff_freq, q_factor."""
    g = torch.Generator().manual_seed(seed)
    gain = torch.rand(B, 1, generator=g) * 40 - 20
    fc = 20 + torch.rand(B, 1, generator=g) ** 2 * 20000
    q = 0.1 + torch.rand(B, 1, generator=g) * 5.9
    wb, wa = torch.randn(B, 3, generator=g), torch.randn(B, 3, generator=g)
    out = dict(gain_db=f32(gain), cutoff_freq=f32(fc), q_factor=f32(q), wb=f32(wb), wa=f32(wa))
    for t in BIQUAD_TYPES:
        for tag, dt in (("32", torch.float32), ("64", torch.float64)):
            ins = [v.to(dt).clone().requires_grad_(True) for v in (gain, fc, q)]
            b, a = dasp_pytorch.signal.biquad(*ins, SR, t)
            ((b * wb.to(dt)).sum() + (a * wa.to(dt)).sum()).backward()
            out[f"{t}_b{tag}"], out[f"{t}_a{tag}"] = f32(b), f32(a)
            # (B, 3): d/d gain_db, cutoff_freq, q_factor; low_pass / high_pass do not depend on gain_db (no gradient: stored as 0)
            out[f"{t}_g{tag}"] = f32(torch.cat([v.grad if v.grad is not None else torch.zeros_like(v) for v in ins], 1))
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, sorted(out))


def lfilter_case(name, B, N, seed):
    """signal.lfilter_via_fsm (dasp_pytorch/signal.py:95-133) on (B, 1, N): the compressor's one-pole smoother (K = 2, :372-380),
    a second-order IIR (K = 3) and an FIR (a = None, K = 3); forward and the gradients w.r.t. x, b, a."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 1, N, generator=g) * 2 - 1
    w = torch.randn(B, 1, N, generator=g)
    alpha = 0.9 + 0.095 * torch.rand(B, 1, generator=g)
    b1p = torch.cat([1 - alpha, torch.zeros(B, 1)], 1)
    a1p = torch.cat([torch.ones(B, 1), -alpha], 1)
    r = 0.5 + 0.45 * torch.rand(B, 1, generator=g); th = 0.1 + 2.9 * torch.rand(B, 1, generator=g)
    a0 = 0.5 + torch.rand(B, 1, generator=g)
    a2 = torch.cat([torch.ones(B, 1), -2 * r * torch.cos(th), r * r], 1) * a0
    b2 = torch.randn(B, 3, generator=g)
    bf = torch.randn(B, 3, generator=g)
    out = dict(x=f32(x), w=f32(w), b_onepole=f32(b1p), a_onepole=f32(a1p), b_iir2=f32(b2), a_iir2=f32(a2), b_fir=f32(bf))
    for key, bb, aa in (("onepole", b1p, a1p), ("iir2", b2, a2), ("fir", bf, None)):
        for tag, dt in (("32", torch.float32), ("64", torch.float64)):
            xx = x.to(dt).clone().requires_grad_(True)
            bt = bb.to(torch.float32).to(dt).clone().requires_grad_(True)
            at = aa.to(torch.float32).to(dt).clone().requires_grad_(True) if aa is not None else None
            y = dasp_pytorch.signal.lfilter_via_fsm(xx, bt, at)
            (y * w.to(dt)).sum().backward()
            out[f"{key}_y{tag}"], out[f"{key}_gx{tag}"], out[f"{key}_gb{tag}"] = f32(y), f32(xx.grad), f32(bt.grad)
            if at is not None:
                out[f"{key}_ga{tag}"] = f32(at.grad)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, sorted(out))


def lfilter_long_case(name, B, N, seed):
    """signal.lfilter_via_fsm (dasp_pytorch/signal.py:95-133) with more than three coefficients, (B, 1, N): a 4th-order Butterworth
    low-pass per item (K = 5, a0 scaled away from 1), a 7th-order Chebyshev-I band of cut-offs (K = 8), a 16-tap FIR (a = None), and one
    6th-order filter shared by the batch (b, a of shape (1, 7)); forward and the gradients w.r.t. x, b, a from the reference's fp32 and
    fp64 runs. The impulse responses have decayed within N, so the reference's frequency-sampling result is the recurrence's."""
    import scipy.signal
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 1, N, generator=g) * 2 - 1
    w = torch.randn(B, 1, N, generator=g)
    cut = 0.08 + 0.5 * torch.rand(B, generator=g)
    a0 = 0.5 + torch.rand(B, 1, generator=g)
    bw4 = [scipy.signal.butter(4, float(c)) for c in cut]
    b5 = torch.tensor(np.stack([q[0] for q in bw4])) * a0
    a5 = torch.tensor(np.stack([q[1] for q in bw4])) * a0
    ch7 = [scipy.signal.cheby1(7, 1.0, float(c)) for c in cut]
    b8 = torch.tensor(np.stack([q[0] for q in ch7]))
    a8 = torch.tensor(np.stack([q[1] for q in ch7]))
    bf = torch.randn(B, 16, generator=g) * 0.3
    sh = scipy.signal.butter(6, 0.3)
    b7 = torch.tensor(sh[0])[None]; a7 = torch.tensor(sh[1])[None] * 1.7
    out = dict(x=f32(x), w=f32(w), b_k5=f32(b5), a_k5=f32(a5), b_k8=f32(b8), a_k8=f32(a8), b_fir16=f32(bf), b_shared7=f32(b7), a_shared7=f32(a7))
    for key, bb, aa in (("k5", b5, a5), ("k8", b8, a8), ("fir16", bf, None), ("shared7", b7, a7)):
        for tag, dt in (("32", torch.float32), ("64", torch.float64)):
            xx = x.to(dt).clone().requires_grad_(True)
            bt = bb.to(torch.float32).to(dt).clone().requires_grad_(True)
            at = aa.to(torch.float32).to(dt).clone().requires_grad_(True) if aa is not None else None
            y = dasp_pytorch.signal.lfilter_via_fsm(xx, bt, at)
            (y * w.to(dt)).sum().backward()
            out[f"{key}_y{tag}"], out[f"{key}_gx{tag}"], out[f"{key}_gb{tag}"] = f32(y), f32(xx.grad), f32(bt.grad)
            if at is not None:
                out[f"{key}_ga{tag}"] = f32(at.grad)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, sorted(out))


def chain_case(name, B, N, seed, noise_seed):
    """BASELINE config 5's effect chain as the reference wires it (examples/style_transfer.py:150-154): ParametricEQ -> Compressor ->
    NoiseShapedReverb -> Gain, each through Processor.process_normalized (dasp_pytorch/modules.py:25-51), mono input (B, 1, N) ->
    stereo output (B, 2, N), default 65,536-sample / 1,023-tap reverb. Stored from the reference's fp32 and fp64 runs: y, grad x and the
    gradients w.r.t. all four normalised parameter tensors (18 + 6 + 25 + 1) - these cross every stage boundary (reverb grad x ->
    compressor grad y -> EQ) - plus the EQ -> compressor prefix output `yec` (the pin for the fused forward kernel). The reverb's noise
    comes from the global CPU generator: `noise_seed` is set immediately before the chain runs (the reverb is its only consumer)."""
    g = torch.Generator().manual_seed(seed)
    mods = [dasp_pytorch.ParametricEQ(SR), dasp_pytorch.Compressor(SR), dasp_pytorch.NoiseShapedReverb(SR), dasp_pytorch.Gain(SR)]
    x = torch.rand(B, 1, N, generator=g) * 2 - 1
    # a slow level envelope so that the compressor's three knee regions are all visited behind the EQ
    env_db = torch.nn.functional.interpolate(torch.rand(B, 1, N // 500 + 2, generator=g) * 40 - 40, size=N, mode="linear")
    x = x * 10 ** (env_db / 20)
    pn = [torch.rand(B, m.num_params, generator=g) for m in mods]
    pn[1][:, 4] = pn[1][:, 4].clamp_min(1e-3 / 12)            # knee_db > 0
    w = torch.randn(B, 2, N, generator=g)
    out = dict(x=f32(x), w=f32(w), noise_seed=np.int64(noise_seed), pn_eq=f32(pn[0]), pn_comp=f32(pn[1]), pn_rev=f32(pn[2]), pn_gain=f32(pn[3]))
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        xx = x.to(dt).clone().requires_grad_(True)
        pp = [p.to(dt).clone().requires_grad_(True) for p in pn]
        torch.manual_seed(noise_seed)
        y = xx.clone()
        y = mods[0].process_normalized(y, pp[0])
        yec = y = mods[1].process_normalized(y, pp[1])
        y = mods[2].process_normalized(y, pp[2])
        y = mods[3].process_normalized(y, pp[3])
        (y * w.to(dt)).sum().backward()
        out["y" + tag], out["gx" + tag], out["yec" + tag] = f32(y), f32(xx.grad), f32(yec)
        for key, p in zip(("eq", "comp", "rev", "gain"), pp):
            out[f"gpn_{key}{tag}"] = f32(p.grad)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lfilter_long":          # round 3 addition; the other files regenerate bit-identically
        lfilter_long_case("lfilter_long_b3_n9000", 3, 9000, seed=131)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "chain":                 # round 4 addition
        torch.set_num_threads(8)
        chain_case("chain_b2c1_n20000", 2, 20000, seed=141, noise_seed=6141)
        sys.exit(0)
    torch.set_num_threads(8)
    eq_case("eq_b3c2_n12000", 3, 2, 12000, 3, seed=101)
    eq_case("eq_bcast_b2c1_n4099", 2, 1, 4099, 1, seed=102)
    sos_case("sos_b2c2_n6000_s3", 2, 2, 6000, 3, seed=103)
    gain_dist_case("gain_dist_cfg1", 4, 1, 16384, seed=104)
    stereo_case("stereo_b2t3_n1501", 2, 3, 1501, seed=110)
    comp_case("comp_b3c2_n12000", 3, 2, 12000, seed=105, speechlike=True)
    comp_case("comp_b2c1_n20011_look7", 2, 1, 20011, seed=106, lookahead=7, speechlike=True)
    reverb_case("rev_b2c2_n6000_l2048_t127", 2, 2, 6000, 2048, 127, seed=107, store_noise=True)
    reverb_case("rev_b1c1_n5000_l1000_t63", 1, 1, 5000, 1000, 63, seed=108, store_noise=True)
    reverb_case("rev_b1c2_n20000_default", 1, 2, 20000, 65536, 1023, seed=109, store_noise=False)
    # round 2: the normalised-parameter API, the coefficient design and the first-order / FIR filter boundary
    norm_case("norm_gain_b3c2_n4000", "Gain", 3, 2, 4000, seed=121)
    norm_case("norm_eq_b3c2_n12000", "ParametricEQ", 3, 2, 12000, seed=122)
    norm_case("norm_comp_b3c2_n12000", "Compressor", 3, 2, 12000, seed=123)
    norm_case("norm_rev_b1c2_n6000", "NoiseShapedReverb", 1, 2, 6000, seed=124, noise_seed=5124)
    biquad_case("biquad_types_b6", 6, seed=125)
    lfilter_case("lfilter_b3_n9000", 3, 9000, seed=126)
    dist_sample_case("dist_sample_b2c2_n3001", 2, 2, 3001, seed=127)
    lfilter_long_case("lfilter_long_b3_n9000", 3, 9000, seed=131)
    chain_case("chain_b2c1_n20000", 2, 20000, seed=141, noise_seed=6141)
