"""CPU oracle: numpy restatement of the reference algorithms on the hot path.

TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg
may import this module; nothing under dasp_pytorch_amd/ does (the product has no CPU path).

Each function restates one reference function (csteinmetz1/dasp-pytorch v0.0.1, file:line cited)
with the *reference's own algorithm* -- e.g. the IIR filters are the frequency-sampling method
(zero-padded FFT, complex divide, inverse FFT, crop), not a recursion. Gradients are hand-derived
vector-Jacobian products of those algorithms (the reference gets them from torch autograd).

Pinning: the reference has no tests or golden vectors, so the pins are outputs
of the reference itself, generated in the build container by tests/golden/make_golden.py (imports
/root/reference) and committed under tests/golden/*.npz; tests/test_oracle_cpu.py checks this module
against every one of them (forward values and gradients, fp32 and fp64).

All functions take/return numpy arrays; `dtype` selects the arithmetic (np.float64 or np.float32;
numpy >= 2 runs float32 FFTs natively, matching the reference's fp32 torch.fft path).
"""
import math

import numpy as np

# ------------------------------------------------------------------------------------------------
# signal.py


def _cplx(dtype):
    return np.complex64 if np.dtype(dtype) == np.float32 else np.complex128


def n_fft_for(T):
    """signal.py:109-110 / :150-151  n_fft = 2 ** ceil(log2(2T - 1))."""
    return int(2 ** math.ceil(math.log2(T + T - 1)))


def fft_freqz(b, a, n_fft):
    """signal.py:7-11."""
    return np.fft.rfft(b, n_fft, axis=-1) / np.fft.rfft(a, n_fft, axis=-1)


def fft_sosfreqz(sos, n_fft):
    """signal.py:14-32. sos (bs, S, 6) -> H (bs, n_fft/2+1)."""
    bs, S, six = sos.shape
    assert six == 6
    H = None
    for s in range(S):
        Hs = fft_freqz(sos[:, s, :3], sos[:, s, 3:], n_fft)
        H = Hs if H is None else H * Hs
    return H


def freqdomain_fir(x, H, n_fft):
    """signal.py:35-39."""
    X = np.fft.rfft(x, n_fft, axis=-1)
    return np.fft.irfft(X * H.astype(X.dtype), n_fft, axis=-1)


def sosfilt_via_fsm(sos, x, dtype=np.float64):
    """signal.py:136-166. sos (bs,S,6), x (bs,...,T) -> y (bs,...,T)."""
    sos = np.asarray(sos, dtype)
    x = np.asarray(x, dtype)
    T = x.shape[-1]
    n_fft = n_fft_for(T)
    H = fft_sosfreqz(sos, n_fft)
    for _ in range(x.ndim - 2):
        H = H[:, None]
    return freqdomain_fir(x, H, n_fft)[..., :T].astype(dtype)


def sosfilt_via_fsm_vjp(sos, x, gy, dtype=np.float64):
    """VJP of sosfilt_via_fsm: returns (gsos (bs,S,6), gx like x).

    y_full = irfft(rfft(x_pad) * prod_s B_s/A_s) is a circular convolution, so
      gx      = irfft(rfft(gy_pad) * conj(H))[:T]
      dL/db_sj =  sum_n gy_pad[n] * q_s[(n-j) mod n_fft],  q_s = irfft(X * H_excl_s / A_s)
      dL/da_sj = -sum_n gy_pad[n] * r_s[(n-j) mod n_fft],  r_s = irfft(X * H / A_s)
    (H_excl_s = product of the other sections), summed over every non-batch dim of x."""
    sos = np.asarray(sos, dtype)
    x = np.asarray(x, dtype)
    gy = np.asarray(gy, dtype)
    bs, S, _ = sos.shape
    T = x.shape[-1]
    n_fft = n_fft_for(T)
    xb = x.reshape(x.shape[0], -1, T)
    gb = gy.reshape(x.shape[0], -1, T)
    Hs = [fft_freqz(sos[:, s, :3], sos[:, s, 3:], n_fft) for s in range(S)]       # (bs, bins)
    As = [np.fft.rfft(sos[:, s, 3:], n_fft, axis=-1) for s in range(S)]
    H = Hs[0].copy()
    for s in range(1, S):
        H = H * Hs[s]
    X = np.fft.rfft(xb, n_fft, axis=-1)
    GY = np.fft.rfft(gb, n_fft, axis=-1)
    gx = np.fft.irfft(GY * np.conj(H)[:, None], n_fft, axis=-1)[..., :T]
    gpad = np.zeros(gb.shape[:-1] + (n_fft,), dtype)
    gpad[..., :T] = gb
    gsos = np.zeros((sos.shape[0], S, 6), np.float64)
    for s in range(S):
        Hex = np.ones_like(H)
        for i in range(S):
            if i != s:
                Hex = Hex * Hs[i]
        q = np.fft.irfft(X * (Hex / As[s])[:, None], n_fft, axis=-1)
        r = np.fft.irfft(X * (H / As[s])[:, None], n_fft, axis=-1)
        for j in range(3):
            gb_j = np.sum(gpad.astype(np.float64) * np.roll(q, j, axis=-1), axis=(1, 2))
            ga_j = -np.sum(gpad.astype(np.float64) * np.roll(r, j, axis=-1), axis=(1, 2))
            if sos.shape[0] == 1 and x.shape[0] != 1:
                gsos[0, s, j] = gb_j.sum()
                gsos[0, s, 3 + j] = ga_j.sum()
            else:
                gsos[:, s, j] = gb_j
                gsos[:, s, 3 + j] = ga_j
    return gsos.astype(dtype), gx.reshape(x.shape).astype(dtype)


def biquad(gain_db, cutoff_freq, q_factor, sample_rate, filter_type="peaking"):
    """signal.py:242-306. Inputs (bs,) -> b, a each (bs, 3), a0-normalised. Works for complex
    inputs too (used for complex-step Jacobians)."""
    A = 10 ** (gain_db / 40.0)
    w0 = 2 * math.pi * (cutoff_freq / sample_rate)
    alpha = np.sin(w0) / (2 * q_factor)
    cos_w0 = np.cos(w0)
    sqrt_A = np.sqrt(A)
    if filter_type == "high_shelf":
        b0 = A * ((A + 1) + (A - 1) * cos_w0 + 2 * sqrt_A * alpha)
        b1 = -2 * A * ((A - 1) + (A + 1) * cos_w0)
        b2 = A * ((A + 1) + (A - 1) * cos_w0 - 2 * sqrt_A * alpha)
        a0 = (A + 1) - (A - 1) * cos_w0 + 2 * sqrt_A * alpha
        a1 = 2 * ((A - 1) - (A + 1) * cos_w0)
        a2 = (A + 1) - (A - 1) * cos_w0 - 2 * sqrt_A * alpha
    elif filter_type == "low_shelf":
        b0 = A * ((A + 1) - (A - 1) * cos_w0 + 2 * sqrt_A * alpha)
        b1 = 2 * A * ((A - 1) - (A + 1) * cos_w0)
        b2 = A * ((A + 1) - (A - 1) * cos_w0 - 2 * sqrt_A * alpha)
        a0 = (A + 1) + (A - 1) * cos_w0 + 2 * sqrt_A * alpha
        a1 = -2 * ((A - 1) + (A + 1) * cos_w0)
        a2 = (A + 1) + (A - 1) * cos_w0 - 2 * sqrt_A * alpha
    elif filter_type == "peaking":
        b0 = 1 + alpha * A
        b1 = -2 * cos_w0
        b2 = 1 - alpha * A
        a0 = 1 + (alpha / A)
        a1 = -2 * cos_w0
        a2 = 1 - (alpha / A)
    elif filter_type == "low_pass":
        b0 = (1 - cos_w0) / 2
        b1 = 1 - cos_w0
        b2 = (1 - cos_w0) / 2
        a0 = 1 + alpha
        a1 = -2 * cos_w0
        a2 = 1 - alpha
    elif filter_type == "high_pass":
        b0 = (1 + cos_w0) / 2
        b1 = -(1 + cos_w0)
        b2 = (1 + cos_w0) / 2
        a0 = 1 + alpha
        a1 = -2 * cos_w0
        a2 = 1 - alpha
    else:
        raise ValueError(f"Invalid filter_type: {filter_type}.")
    b = np.stack([b0, b1, b2], -1) / a0[..., None]
    a = np.stack([a0, a1, a2], -1) / a0[..., None]
    return b, a


# ------------------------------------------------------------------------------------------------
# functional.py

PEQ_SECTIONS = [("low_shelf", "low_shelf"), ("band0", "peaking"), ("band1", "peaking"), ("band2", "peaking"),
                ("band3", "peaking"), ("high_shelf", "high_shelf")]


def biquad_vjp(gain_db, cutoff_freq, q_factor, sample_rate, filter_type, gb, ga):
    """Gradient of sum(b * gb) + sum(a * ga) w.r.t. (gain_db, cutoff_freq, q_factor), (bs, 3), by fp64 central differences of `biquad`
    (the reference gets it from autograd through signal.py:255-304; the design is a smooth closed form, so differences at a relative
    step of 1e-6 are good to ~1e-9)."""
    ins = [np.asarray(v, np.float64).reshape(-1) for v in (gain_db, cutoff_freq, q_factor)]
    out = np.zeros((ins[0].shape[0], 3))
    for d in range(3):
        h = 1e-6 * np.maximum(1.0, np.abs(ins[d]))
        vals = []
        for sgn in (+1.0, -1.0):
            mod = [v.copy() for v in ins]
            mod[d] = mod[d] + sgn * h
            b, a = biquad(*mod, sample_rate, filter_type)
            vals.append(np.sum(b * gb, 1) + np.sum(a * ga, 1))
        out[:, d] = (vals[0] - vals[1]) / (2 * h)
    return out


def peq_sos(params, sample_rate, dtype=np.float64):
    """functional.py:211-265: params (bs, 18) in the reference's argument order -> sos (bs,6,6)."""
    params = np.asarray(params)
    bs = params.shape[0]
    sos = np.zeros((bs, 6, 6), dtype if not np.iscomplexobj(params) else np.complex128)
    for k, (_, ftype) in enumerate(PEQ_SECTIONS):
        p = params[:, 3 * k:3 * k + 3].astype(sos.dtype)
        b, a = biquad(p[:, 0], p[:, 1], p[:, 2], sample_rate, ftype)
        sos[:, k, :3] = b
        sos[:, k, 3:] = a
    return sos


def parametric_eq(x, sample_rate, params, dtype=np.float64):
    """functional.py:118-272 with the 18 controls stacked as params (bs or 1, 18)."""
    sos = peq_sos(np.asarray(params, dtype), sample_rate, dtype)
    if sos.shape[0] == 1 and x.shape[0] != 1:
        sos = np.repeat(sos, x.shape[0], 0)
    return sosfilt_via_fsm(sos, x, dtype)


def parametric_eq_vjp(x, sample_rate, params, gy, dtype=np.float64):
    """Returns (gx, gparams (bs or 1, 18)). d(sos)/d(params) by complex-step differentiation of
    `biquad` (analytic formulas), which is exact to rounding."""
    params = np.asarray(params, np.float64)
    bp = params.shape[0]
    sos = peq_sos(params.astype(dtype), sample_rate, dtype)
    sos_b = np.repeat(sos, x.shape[0], 0) if (bp == 1 and x.shape[0] != 1) else sos
    gsos, gx = sosfilt_via_fsm_vjp(sos_b, x, gy, dtype)
    if bp == 1 and x.shape[0] != 1:
        gsos = gsos.sum(0, keepdims=True)
    gparams = np.zeros((bp, 18))
    h = 1e-30
    for i in range(18):
        pc = params.astype(np.complex128)
        pc[:, i] += 1j * h
        dsos = peq_sos(pc, sample_rate).imag / h          # (bp, 6, 6)
        gparams[:, i] = np.sum(gsos.astype(np.float64) * dsos, axis=(1, 2))
    return gx, gparams.astype(dtype)


def gain(x, sample_rate, gain_db, dtype=np.float64):
    """functional.py:10-29."""
    x = np.asarray(x, dtype)
    g = np.asarray(gain_db, dtype).reshape(x.shape[0], 1, 1)
    return x * (10 ** (np.repeat(g, x.shape[1], 1) / dtype(20.0)))


def gain_vjp(x, sample_rate, gain_db, gy, dtype=np.float64):
    x = np.asarray(x, dtype)
    gy = np.asarray(gy, dtype)
    g = np.asarray(gain_db, dtype).reshape(x.shape[0], 1, 1)
    lin = 10 ** (g / 20.0)
    gx = gy * lin
    ggain = np.sum(gy * x * lin, axis=(1, 2)) * (math.log(10.0) / 20.0)
    return gx.astype(dtype), ggain.astype(dtype)


def stereo_widener(x, sample_rate, width, dtype=np.float64):
    """functional.py:580-605: mid/side, mid *= 2 (1 - width), side *= 2 width, back to left/right. x (bs, 2, N), width (bs)."""
    x = np.asarray(x, dtype)
    w = np.asarray(width, dtype).reshape(-1, 1)
    sqrt2 = dtype(np.sqrt(2.0))
    mid = (x[:, 0] + x[:, 1]) / sqrt2 * (2 * (1 - w))
    side = (x[:, 0] - x[:, 1]) / sqrt2 * (2 * w)
    return np.stack(((mid + side) / sqrt2, (mid - side) / sqrt2), 1)


def stereo_widener_vjp(x, sample_rate, width, gy, dtype=np.float64):
    x = np.asarray(x, dtype)
    gy = np.asarray(gy, dtype)
    k = 1 - 2 * np.asarray(width, dtype).reshape(-1, 1)          # left = L + k R, right = k L + R
    gx = np.stack((gy[:, 0] + k * gy[:, 1], k * gy[:, 0] + gy[:, 1]), 1)
    gw = -2 * np.sum(x[:, 1] * gy[:, 0] + x[:, 0] * gy[:, 1], axis=1)
    return gx.astype(dtype), gw.reshape(np.asarray(width).shape).astype(dtype)


def _pan_gains(pan, dtype):
    """functional.py:621-626."""
    theta = np.asarray(pan, dtype) * (np.pi / 2)
    return np.sqrt(((np.pi / 2) - theta) * (2 / np.pi) * np.cos(theta)), np.sqrt(theta * (2 / np.pi) * np.sin(theta)), theta


def stereo_panner(x, sample_rate, pan, dtype=np.float64):
    """functional.py:608-636: x (bs, T, N), pan (bs, T) -> (bs, 2, T, N) (what the code returns; the docstring says (bs, T, 2, N))."""
    x = np.asarray(x, dtype)
    lg, rg, _ = _pan_gains(np.asarray(pan).reshape(x.shape[0], x.shape[1]), dtype)
    return np.stack((x * lg[..., None], x * rg[..., None]), 1)


def stereo_panner_vjp(x, sample_rate, pan, gy, dtype=np.float64):
    x = np.asarray(x, dtype)
    gy = np.asarray(gy, dtype)
    lg, rg, th = _pan_gains(np.asarray(pan).reshape(x.shape[0], x.shape[1]), dtype)
    gx = gy[:, 0] * lg[..., None] + gy[:, 1] * rg[..., None]
    dlg = 0.5 / lg * (2 / np.pi) * (-np.cos(th) - (np.pi / 2 - th) * np.sin(th)) * (np.pi / 2)
    drg = 0.5 / rg * (2 / np.pi) * (np.sin(th) + th * np.cos(th)) * (np.pi / 2)
    gp = dlg * np.sum(gy[:, 0] * x, -1) + drg * np.sum(gy[:, 1] * x, -1)
    return gx.astype(dtype), gp.reshape(np.asarray(pan).shape).astype(dtype)


def stereo_bus(x, sample_rate, send_db, dtype=np.float64):
    """functional.py:32-62: x (bs, 2, T, N), send_db (bs, T, 1) -> (bs, 2, N)."""
    x = np.asarray(x, dtype)
    s = 10 ** (np.asarray(send_db, dtype).reshape(x.shape[0], 1, x.shape[2], 1) / dtype(20.0))
    return np.sum(x * s, axis=2)


def stereo_bus_vjp(x, sample_rate, send_db, gy, dtype=np.float64):
    x = np.asarray(x, dtype)
    gy = np.asarray(gy, dtype)
    s = 10 ** (np.asarray(send_db, dtype).reshape(x.shape[0], 1, x.shape[2], 1) / 20.0)
    gx = gy[:, :, None, :] * s
    gs = np.sum(gy[:, :, None, :] * x, axis=(1, 3)) * s[:, 0, :, 0] * (math.log(10.0) / 20.0)
    return gx.astype(dtype), gs.reshape(np.asarray(send_db).shape).astype(dtype)


def distortion(x, sample_rate, drive_db, dtype=np.float64):
    """functional.py:65-78: drive_db.view(bs, chs, -1) -> bs*chs drive values (one per row) or bs*chs*seq_len (one per sample)."""
    x = np.asarray(x, dtype)
    d = np.asarray(drive_db, dtype).reshape(x.shape[0], x.shape[1], -1)
    return np.tanh(x * (10 ** (d / dtype(20.0))))


def distortion_vjp(x, sample_rate, drive_db, gy, dtype=np.float64):
    x = np.asarray(x, dtype)
    gy = np.asarray(gy, dtype)
    d = np.asarray(drive_db, dtype).reshape(x.shape[0], x.shape[1], -1)
    lin = 10 ** (d / 20.0)
    y = np.tanh(x * lin)
    t = gy * (1 - y * y)
    gx = t * lin
    gd = t * x * lin * (math.log(10.0) / 20.0)
    if d.shape[2] == 1:                      # one drive per row: the broadcast's adjoint sums over time
        gd = np.sum(gd, axis=2, keepdims=True)
    return gx.astype(dtype), gd.reshape(np.asarray(drive_db).shape).astype(dtype)


# ------------------------------------------------------------------------------------------------
# compressor  (functional.py:275-399, smoothing filter through signal.py:95-133)


def lfilter_via_fsm(x, b, a=None, dtype=np.float64):
    """signal.py:95-133. x (bs,1,T), b, a (bs,K) -> y (bs,1,T); a = None: FIR (H = rfft(b), :115-117)."""
    x = np.asarray(x, dtype)
    T = x.shape[-1]
    n_fft = n_fft_for(T)
    b = np.asarray(b, dtype)
    H = np.fft.rfft(b, n_fft, axis=-1) if a is None else fft_freqz(b, np.asarray(a, dtype), n_fft)
    return freqdomain_fir(x, H[:, None], n_fft)[..., :T].astype(dtype)


def lfilter_via_fsm_vjp(x, b, a, gy, dtype=np.float64):
    """VJP of lfilter_via_fsm: (gx, gb, ga or None). With X = rfft(x_pad), H = B / A (A = 1 for an FIR):
      gx = irfft(rfft(gy_pad) conj(H))[:T];  dL/db_j = sum_n gy_pad[n] q[(n - j) mod n_fft], q = irfft(X / A);
      dL/da_j = -sum_n gy_pad[n] r[(n - j) mod n_fft], r = irfft(X H / A)."""
    x, gy, b = np.asarray(x, dtype), np.asarray(gy, dtype), np.asarray(b, dtype)
    bs, _, T = x.shape
    K = b.shape[-1]
    n_fft = n_fft_for(T)
    Bf = np.fft.rfft(b, n_fft, axis=-1)
    Af = np.fft.rfft(np.asarray(a, dtype), n_fft, axis=-1) if a is not None else np.ones_like(Bf)
    H = Bf / Af
    X = np.fft.rfft(x[:, 0], n_fft, axis=-1)
    GY = np.fft.rfft(gy[:, 0], n_fft, axis=-1)
    gx = np.fft.irfft(GY * np.conj(H), n_fft, axis=-1)[:, None, :T]
    gpad = np.zeros((bs, n_fft), np.float64)
    gpad[:, :T] = gy[:, 0]
    q = np.fft.irfft(X / Af, n_fft, axis=-1)
    r = np.fft.irfft(X * H / Af, n_fft, axis=-1)
    gb = np.stack([np.sum(gpad * np.roll(q, j, axis=-1), axis=-1) for j in range(K)], 1)
    ga = -np.stack([np.sum(gpad * np.roll(r, j, axis=-1), axis=-1) for j in range(K)], 1) if a is not None else None
    return gx.astype(dtype), gb.astype(dtype), (ga.astype(dtype) if ga is not None else None)


def _compressor_core(x, sample_rate, threshold_db, ratio, attack_ms, knee_db, makeup_gain_db, eps, lookahead, dtype):
    x = np.asarray(x, dtype)
    bs, chs, T = x.shape
    col = lambda v: np.asarray(v, dtype).reshape(bs, 1, 1)
    thr, rat, atk, knee, mk = col(threshold_db), col(ratio), col(attack_ms), col(knee_db), col(makeup_gain_db)
    x_side = x.sum(1, keepdims=True)                                     # :328
    nat = sample_rate * (atk / dtype(1e3))                               # :339
    alpha = np.exp(-np.log(dtype(9.0)) / nat)                            # :341-342
    mag = np.maximum(np.abs(x_side), dtype(eps))
    x_db = 20 * np.log10(mag)                                            # :347
    lo, hi = thr - knee / 2, thr + knee / 2
    in_knee = (x_db >= lo) & (x_db <= hi)                                # :355-357
    above = x_db > hi                                                    # :364
    with np.errstate(divide="ignore", invalid="ignore"):
        x_sc_knee = x_db + ((1 / rat) - 1) * ((x_db - thr + knee / 2) ** 2) / (2 * knee)   # :358-360
    x_sc = np.where(in_knee, x_sc_knee, x_db)
    x_sc = np.where(above, thr + (x_db - thr) / rat, x_sc)               # :365-366
    g_c = x_sc - x_db                                                    # :369
    b = np.concatenate([1 - alpha, np.zeros_like(alpha)], -1)[:, 0]      # :372-375
    a = np.concatenate([np.ones_like(alpha), -alpha], -1)[:, 0]          # :376-379
    g = lfilter_via_fsm(g_c, b, a, dtype)                                # :380
    x_d = x
    if lookahead > 0:                                                    # :383-385
        x_d = np.roll(x, lookahead, axis=-1)
        x_d[:, :, :lookahead] = 0
    lin = 10 ** ((g + mk) / dtype(20.0))                                 # :388-391
    return dict(x=x, x_side=x_side, x_db=x_db, mag=mag, in_knee=in_knee, above=above, g_c=g_c, g=g, lin=lin, x_d=x_d,
                thr=thr, rat=rat, atk=atk, knee=knee, mk=mk, alpha=alpha, nat=nat, b=b, a=a)


def compressor(x, sample_rate, threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db, eps=1e-8,
               lookahead_samples=0, dtype=np.float64):
    """functional.py:275-399. release_ms is accepted and unused, exactly like the reference (:340,343-344)."""
    c = _compressor_core(x, sample_rate, threshold_db, ratio, attack_ms, knee_db, makeup_gain_db, eps, lookahead_samples, dtype)
    return (c["x_d"] * c["lin"]).astype(dtype)                            # :394


def compressor_vjp(x, sample_rate, threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db, gy, eps=1e-8,
                   lookahead_samples=0, dtype=np.float64):
    """Hand-derived VJP of `compressor` (the reference uses autograd through the FFT filter).
    Returns (gx, dict of control grads: threshold_db, ratio, attack_ms, release_ms (zeros), knee_db, makeup_gain_db)."""
    c = _compressor_core(x, sample_rate, threshold_db, ratio, attack_ms, knee_db, makeup_gain_db, eps, lookahead_samples, dtype)
    gy = np.asarray(gy, dtype)
    bs, chs, T = c["x"].shape
    k = lookahead_samples
    lin, x_d = c["lin"], c["x_d"]
    gxd = gy * lin                                                        # d/dx_d
    gx = gxd.copy()
    if k > 0:
        gx = np.zeros_like(gxd)
        gx[:, :, :T - k] = gxd[:, :, k:]
    ggs = np.sum(gy * x_d, 1, keepdims=True) * lin * (math.log(10.0) / 20.0)    # dL/d(g + makeup)
    g_mk = ggs.sum((1, 2))
    # adjoint of the FSM filter (circular convolution on n_fft points) wrt its input and coefficients
    n_fft = n_fft_for(T)
    b, a = c["b"], c["a"]
    Bf, Af = np.fft.rfft(b, n_fft, axis=-1), np.fft.rfft(a, n_fft, axis=-1)
    H = Bf / Af
    G = np.fft.rfft(ggs, n_fft, axis=-1)
    g_gc = np.fft.irfft(G * np.conj(H)[:, None], n_fft, axis=-1)[..., :T]
    X = np.fft.rfft(c["g_c"], n_fft, axis=-1)
    gpad = np.zeros((bs, 1, n_fft), np.float64)
    gpad[..., :T] = ggs
    q = np.fft.irfft(X / Af[:, None], n_fft, axis=-1)
    r = np.fft.irfft(X * (H / Af)[:, None], n_fft, axis=-1)
    g_b0 = np.sum(gpad * q, (1, 2))
    g_a1 = -np.sum(gpad * np.roll(r, 1, -1), (1, 2))
    g_alpha = -g_b0 - g_a1                                                # b0 = 1 - alpha, a1 = -alpha
    alpha, nat, atk = c["alpha"][:, 0, 0], c["nat"][:, 0, 0], c["atk"][:, 0, 0]
    g_atk = g_alpha * alpha * math.log(9.0) / (nat * nat) * (sample_rate / 1e3)
    # gain computer
    thr, rat, knee, x_db = c["thr"], c["rat"], c["knee"], c["x_db"]
    qk = x_db - thr + knee / 2
    ik, ab = c["in_knee"], c["above"]
    with np.errstate(divide="ignore", invalid="ignore"):
        d_xdb = np.where(ab, 1 / rat - 1, np.where(ik, (1 / rat - 1) * qk / knee, 0.0))
        d_thr = np.where(ab, 1 - 1 / rat, np.where(ik, -(1 / rat - 1) * qk / knee, 0.0))
        d_rat = np.where(ab, -(x_db - thr) / rat ** 2, np.where(ik, -(qk ** 2) / (2 * knee) / rat ** 2, 0.0))
        d_knee = np.where(ik, (1 / rat - 1) * (qk / (2 * knee) - qk ** 2 / (2 * knee ** 2)), 0.0)
    g_thr = np.sum(g_gc * d_thr, (1, 2))
    g_rat = np.sum(g_gc * d_rat, (1, 2))
    g_knee = np.sum(g_gc * d_knee, (1, 2))
    g_xdb = g_gc * d_xdb
    s = c["x_side"]
    g_side = np.where(np.abs(s) >= eps, g_xdb * (20.0 / math.log(10.0)) * np.sign(s) / c["mag"], 0.0)   # clamp passes grad at >= eps
    gx = gx + g_side                                                      # x_side = x.sum(1)
    shp = np.asarray(threshold_db).shape
    grads = dict(threshold_db=g_thr, ratio=g_rat, attack_ms=g_atk, release_ms=np.zeros(bs), knee_db=g_knee, makeup_gain_db=g_mk)
    return gx.astype(dtype), {k_: v.reshape(shp).astype(dtype) for k_, v in grads.items()}


# ------------------------------------------------------------------------------------------------
# expander -- PARITY UNPINNED: the reference's expander() is a stub (functional.py:402-403), so this
# is a statement of the design the HIP kernel implements (mode 1 of csrc/dynamics.hip), not of any
# reference behaviour. Same structure as compressor(); only the static curve differs.


def expander(x, sample_rate, threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db, eps=1e-8,
             lookahead_samples=0, dtype=np.float64):
    x = np.asarray(x, dtype)
    bs, chs, T = x.shape
    col = lambda v: np.asarray(v, dtype).reshape(bs, 1, 1)
    thr, rat, atk, knee, mk = col(threshold_db), col(ratio), col(attack_ms), col(knee_db), col(makeup_gain_db)
    x_side = x.sum(1, keepdims=True)
    alpha = np.exp(-np.log(dtype(9.0)) / (sample_rate * (atk / dtype(1e3))))
    x_db = 20 * np.log10(np.maximum(np.abs(x_side), dtype(eps)))
    lo, hi = thr - knee / 2, thr + knee / 2
    in_knee = (x_db >= lo) & (x_db <= hi) & (knee > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        x_sc = np.where(x_db < lo, thr + (x_db - thr) * rat, np.where(in_knee, x_db + (1 - rat) * (x_db - hi) ** 2 / (2 * knee), x_db))
    g_c = x_sc - x_db
    from scipy.signal import lfilter
    g = np.stack([lfilter([1 - alpha[b, 0, 0]], [1.0, -alpha[b, 0, 0]], g_c[b], axis=-1) for b in range(bs)])
    x_d = x
    if lookahead_samples > 0:
        x_d = np.roll(x, lookahead_samples, axis=-1)
        x_d[:, :, :lookahead_samples] = 0
    return (x_d * 10 ** ((g + mk) / dtype(20.0))).astype(dtype)


# ------------------------------------------------------------------------------------------------
# noise_shaped_reverberation  (functional.py:406-577, filterbank signal.py:42-92)
# The reference evaluates both convolutions directly (conv1d); the restatement uses SciPy's fp64
# FFT convolution, which computes the same linear convolutions to rounding.

OCTAVE_BANDS = [31.5, 63, 125, 250, 500, 1000, 2000, 4000, 8000, 16000]


def octave_band_filterbank(num_taps, sample_rate):
    """signal.py:42-92 -> (12, num_taps) float32 (the reference flips each symmetric filter: a no-op)."""
    from scipy.signal import firwin
    filts = [firwin(num_taps, 12, fs=sample_rate)]                                    # :60-64
    for fc in OCTAVE_BANDS:                                                          # :69-80
        f_min, f_max = fc / np.sqrt(2), np.clip(fc * np.sqrt(2), 0, (sample_rate / 2) * 0.999)
        filts.append(firwin(num_taps, [f_min, f_max], fs=sample_rate, pass_zero=False))
    filts.append(firwin(num_taps, 18000, fs=sample_rate, pass_zero=False))            # :84
    return np.stack([f.astype(np.float32)[::-1] for f in filts], 0)


def _reverb_core(x, sample_rate, gains, decays, mix, noise, num_samples, num_bandpass_taps, dtype):
    from scipy.signal import fftconvolve
    x = np.asarray(x, dtype)
    bs, chs, T = x.shape
    assert chs <= 2 and num_bandpass_taps % 2 == 1                                    # :487,490
    if chs == 1:
        x = np.repeat(x, 2, 1)                                                        # :493-495
    L = num_samples
    filt = octave_band_filterbank(num_bandpass_taps, sample_rate).astype(dtype)       # :537-538
    nb = filt.shape[0]
    g = np.asarray(gains, dtype).reshape(bs, 1, nb, 1)
    d = np.asarray(decays, dtype).reshape(bs, 1, nb, 1) * 10.0 + 1.0                  # :562
    m = np.asarray(mix, dtype).reshape(bs, 1, 1)
    wn = np.asarray(noise, dtype)                                                     # (2 bs, nb, L + taps - 1), :548
    # conv1d = cross-correlation with the stored filter, "valid" (:551-556)
    wf = fftconvolve(wn, filt[None, :, ::-1], mode="valid", axes=-1).reshape(bs, 2, nb, L)
    t = np.linspace(0, 1, L).astype(dtype)                                            # :561
    env = np.exp(-d * t.reshape(1, 1, 1, -1))                                         # :563
    ir = (wf * env * g).mean(2)                                                       # :564-567  (bs, 2, L)
    y_wet = fftconvolve(x, ir, mode="full", axes=-1)[..., :T]                         # :570-572 causal, truncated
    return dict(x=x, wf=wf, env=env, g=g, d=d, m=m, t=t, ir=ir, y_wet=y_wet, nb=nb, L=L, T=T, chs=chs)


def noise_shaped_reverberation(x, sample_rate, gains, decays, mix, noise, num_samples=65536, num_bandpass_taps=1023, dtype=np.float64):
    """functional.py:406-577 with band gains/decays stacked as (bs, 12) and the white noise passed in explicitly
    (the reference draws torch.randn(bs*2, 12, num_samples + taps - 1) from the global CPU generator, :548)."""
    c = _reverb_core(x, sample_rate, gains, decays, mix, noise, num_samples, num_bandpass_taps, dtype)
    return ((1 - c["m"]) * c["x"] + c["m"] * c["y_wet"]).astype(dtype)                # :575


def noise_shaped_reverberation_vjp(x, sample_rate, gains, decays, mix, noise, gy, num_samples=65536, num_bandpass_taps=1023,
                                   dtype=np.float64):
    """Returns gx (shape of x), ggains (bs,12), gdecays (bs,12), gmix (bs,)."""
    from scipy.signal import fftconvolve
    c = _reverb_core(x, sample_rate, gains, decays, mix, noise, num_samples, num_bandpass_taps, dtype)
    gy = np.asarray(gy, dtype)
    T, L = c["T"], c["L"]
    gwet = c["m"] * gy
    # y_wet[n] = sum_j ir[j] x[n-j]  ->  gx[m] = sum_j ir[j] gwet[m+j] ;  gir[j] = sum_n gwet[n] x[n-j]
    gx = (1 - c["m"]) * gy + fftconvolve(gwet, c["ir"][..., ::-1], mode="full", axes=-1)[..., L - 1:L - 1 + T]
    full = fftconvolve(gwet, c["x"][..., ::-1], mode="full", axes=-1)                  # lag j at index T-1+j
    gir = np.zeros_like(c["ir"])
    nlag = min(L, T)
    gir[..., :nlag] = full[..., T - 1:T - 1 + nlag]
    gmix = np.sum(gy * (c["y_wet"] - c["x"]), (1, 2))
    w = gir[:, :, None, :] * c["wf"] * c["env"] / c["nb"]                             # (bs, 2, nb, L)
    ggain = w.sum((1, 3))
    gdec = (w * c["g"] * (-10.0 * c["t"].reshape(1, 1, 1, -1))).sum((1, 3))
    if c["chs"] == 1:
        gx = gx.sum(1, keepdims=True)
    return gx.astype(dtype), ggain.astype(dtype), gdec.astype(dtype), gmix.astype(dtype)


# ------------------------------------------------------------------------------------------------
# Multi-resolution STFT loss: the op downstream of the hot path in the reference's training loops
# (auraloss.freq.MultiResolutionSTFTLoss(), call sites examples/style_transfer.py:341,363, auto_eq.py:252, virtual_analog.py:288).
# auraloss is a third-party dependency that is NOT vendored in /root/reference and not pinned by it (no version in setup.py /
# pyproject.toml): PARITY UNPINNED. This restates the published algorithm of auraloss 0.4.0 with its default arguments:
#   for (n_fft, hop, win) in ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)):
#       X = torch.stft(x, n_fft, hop, win, hann_window(win), return_complex=True)        # center=True, reflect padding, onesided
#       mag = sqrt(clamp(re^2 + im^2, min=eps)), eps = 1e-8
#       sc = ||mag_target - mag_input||_F / ||mag_target||_F;  lm = mean |log mag_input - log mag_target|
#   loss = mean over resolutions of (sc + lm)
MRSTFT_DEFAULT = ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240))


def _stft_mag(x, n_fft, hop, win, eps, dtype):
    """x (rows, N) -> magnitudes (rows, n_fft/2+1, frames), torch.stft conventions (center, reflect, periodic hann zero-padded to n_fft)."""
    x = np.asarray(x, dtype)
    rows, N = x.shape
    xp = np.pad(x, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
    w = np.zeros(n_fft, dtype)
    lp = (n_fft - win) // 2
    w[lp:lp + win] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win) / win)
    frames = 1 + N // hop
    idx = np.arange(frames)[:, None] * hop + np.arange(n_fft)[None, :]
    seg = xp[:, idx] * w                                   # (rows, frames, n_fft)
    X = np.fft.rfft(seg, axis=-1)
    return np.sqrt(np.maximum(X.real ** 2 + X.imag ** 2, eps)).transpose(0, 2, 1), seg, w, idx


def mrstft_loss(inp, target, resolutions=MRSTFT_DEFAULT, eps=1e-8, dtype=np.float64):
    """inp, target (bs, chs, N) -> scalar loss."""
    a = np.asarray(inp, dtype).reshape(-1, np.shape(inp)[-1]); b = np.asarray(target, dtype).reshape(-1, np.shape(target)[-1])
    total = 0.0
    for n_fft, hop, win in resolutions:
        A, _, _, _ = _stft_mag(a, n_fft, hop, win, eps, dtype)
        B, _, _, _ = _stft_mag(b, n_fft, hop, win, eps, dtype)
        total += np.linalg.norm(B - A) / np.linalg.norm(B) + np.mean(np.abs(np.log(A) - np.log(B)))
    return total / len(resolutions)


def mrstft_loss_vjp(inp, target, resolutions=MRSTFT_DEFAULT, eps=1e-8, dtype=np.float64):
    """d loss / d inp (same shape as inp); the target is treated as a constant (it is the reference signal at every call site)."""
    shape = np.shape(inp)
    a = np.asarray(inp, dtype).reshape(-1, shape[-1]); b = np.asarray(target, dtype).reshape(-1, shape[-1])
    rows, N = a.shape
    g = np.zeros_like(a)
    for n_fft, hop, win in resolutions:
        B, _, _, _ = _stft_mag(b, n_fft, hop, win, eps, dtype)
        A, seg, w, idx = _stft_mag(a, n_fft, hop, win, eps, dtype)
        X = np.fft.rfft(seg, axis=-1).transpose(0, 2, 1)                  # (rows, bins, frames)
        s1, s2 = np.linalg.norm(B - A), np.linalg.norm(B)
        gA = (A - B) / (s1 * s2) - np.sign(np.log(B) - np.log(A)) / (A * A.size)
        gA = np.where(X.real ** 2 + X.imag ** 2 > eps, gA, 0.0) / len(resolutions)     # clamp has zero slope
        G = (gA * X / A).transpose(0, 2, 1)                                # gradient w.r.t. (Re, Im) of the one-sided bins, as complex
        full = np.zeros((rows, G.shape[1], n_fft), np.complex128)
        full[:, :, :n_fft // 2 + 1] = G
        gseg = np.real(np.fft.ifft(full, axis=-1) * n_fft) * w             # sum_k G[k] e^{+i theta}
        gp = np.zeros((rows, N + 2 * (n_fft // 2)), dtype)
        np.add.at(gp, (np.arange(rows)[:, None, None], idx[None, :, :]), gseg)
        # adjoint of the reflect padding
        p = n_fft // 2
        core = gp[:, p:p + N].copy()
        core[:, 1:p + 1] += gp[:, :p][:, ::-1]
        core[:, N - 1 - p:N - 1] += gp[:, p + N:][:, ::-1]
        g += core
    return g.reshape(shape)
