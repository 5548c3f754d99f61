"""DSP primitives with the reference's names and signatures (dasp_pytorch/signal.py).

`sosfilt_via_fsm` keeps the reference's name for drop-in use, but it is evaluated as an exact
recurrence (chunked parallel scan, csrc/sosfilt.hip) instead of the reference's frequency-sampling
approximation; the two agree to <= 1e-13 in fp64 for stable filters whose impulse response has
decayed within the signal length. float64 input takes the double-precision
kernels of csrc/ref64.hip (ops64.py): float64 in, float64 arithmetic, as in the reference.
"""
import functools

import numpy as np
import scipy.signal
import torch

from . import ops as _ops
from .ops import FILTER_TYPES, BiquadFunction, SosFiltFunction
from .ops64 import LFilterFunction, SosFilt64Function, is_f64


def biquad(gain_db: torch.Tensor, cutoff_freq: torch.Tensor, q_factor: torch.Tensor, sample_rate: float, filter_type: str = "peaking"):
    """RBJ-cookbook biquad design (reference: dasp_pytorch/signal.py:242-306): gain_db, cutoff_freq, q_factor with bs values each
    (the reference's (bs, 1)) -> b, a of shape (bs, 3), normalised by a0 (so a[:, 0] == 1), dtype and device of gain_db.
    filter_type: "peaking", "low_shelf", "high_shelf", "low_pass", "high_pass"; anything else raises ValueError as in the reference.
    The design runs on the device in fp64 (the kernel the fused parametric_eq path uses, csrc/sosfilt.hip rbj_design) and is
    differentiable w.r.t. all three controls through its in-kernel Jacobian."""
    if filter_type not in FILTER_TYPES:
        raise ValueError(f"Invalid filter_type: {filter_type}.")
    bs = gain_db.size(0)
    if any(t.numel() != bs for t in (gain_db, cutoff_freq, q_factor)):
        raise RuntimeError(f"biquad: gain_db, cutoff_freq and q_factor must hold one value per batch item ({bs}); "
                           f"got {[tuple(t.shape) for t in (gain_db, cutoff_freq, q_factor)]}")
    ba = BiquadFunction.apply(gain_db.reshape(bs), cutoff_freq.reshape(bs), q_factor.reshape(bs), float(sample_rate), FILTER_TYPES[filter_type])
    ba = ba.to(gain_db.dtype)
    return ba[:, :3], ba[:, 3:]


def lfilter_via_fsm(x: torch.Tensor, b: torch.Tensor, a: torch.Tensor = None):
    """IIR / FIR filter along the last dimension of x (reference: dasp_pytorch/signal.py:95-133): x (bs, 1, timesteps), b (bs, K)
    numerator and a (bs, K) denominator coefficients (any a0), or a = None for an FIR filter. As in the reference x must have one
    channel (`assert chs == 1`). Evaluated as an exact recurrence - one section of the cascaded-biquad scan (csrc/sosfilt.hip) - instead
    of the reference's frequency-sampling approximation; differentiable w.r.t. x, b and a.
    K <= 3 (the reference's only caller is the compressor's one-pole smoother, K = 2, functional.py:372-380) is one section of the
    cascaded-biquad kernels. K = 4 .. 16 runs the recurrence of order K - 1 in double arithmetic, chunks of time side by side and
    stitched by their state transition (csrc/lfilter.hip - the boundary's long tail, not one of the tuned kernels). More than 16
    coefficients raise NotImplementedError (factor into second-order sections and call sosfilt_via_fsm)."""
    bs, chs, seq_len = x.size()  # enforce shape
    assert chs == 1
    K = b.shape[-1]
    if b.dim() != 2 or b.shape[0] not in (1, bs) or (a is not None and a.shape != b.shape):
        raise RuntimeError(f"lfilter_via_fsm: b (and a) must have shape ({bs}, K); got {tuple(b.shape)}" + (f", {tuple(a.shape)}" if a is not None else ""))
    if K > 16:
        raise NotImplementedError(f"lfilter_via_fsm: K = {K} coefficients; recurrences of up to 16 coefficients are evaluated directly. "
                                  "Factor the filter into second-order sections, e.g. sos = scipy.signal.tf2sos(b, a) per batch item, "
                                  "stack them as (bs, n_sections, 6) and call dasp_pytorch_amd.signal.sosfilt_via_fsm(sos, x) "
                                  "(differentiable w.r.t. the sections; any number of sections)")
    if K > 3:
        b = b.type_as(x).double()                        # the reference's rounding of the coefficients (signal.py:113,119), then exact
        if a is None:
            an = torch.zeros_like(b)
            an[:, 0] = 1.0
            bn = b
        else:
            a = a.type_as(x).double()
            bn, an = b / a[:, :1], a / a[:, :1]          # H = B / A for any a0 (signal.py:7-11); torch differentiates the normalisation
        return LFilterFunction.apply(x, bn, an)
    b = b.type_as(x)
    if a is None:
        a = torch.zeros_like(b)
        a[:, 0] = 1.0
    else:
        a = a.type_as(x)
    if K < 3:
        pad = torch.zeros(b.shape[0], 3 - K, dtype=b.dtype, device=b.device)
        b, a = torch.cat([b, pad], 1), torch.cat([a, pad], 1)
    sos = torch.cat([b, a], 1).unsqueeze(1)                  # (bs, 1, 6) rows [b0 b1 b2 a0 a1 a2]
    return (SosFilt64Function if is_f64(x) else SosFiltFunction).apply(sos, x)


def _frequency_domain_helper(name, line):
    def fn(*args, **kwargs):
        raise NotImplementedError(
            f"dasp_pytorch_amd.signal.{name}: the reference's frequency-sampling internals (dasp_pytorch/signal.py:{line}) have no counterpart "
            "here - the filters are evaluated as exact recurrences (sosfilt_via_fsm, lfilter_via_fsm), not as spectra")
    fn.__name__ = name
    return fn


# The reference's L1 helpers of the frequency-sampling method are not part of this package's boundary; they exist as names
# so that `from dasp_pytorch_amd.signal import *` fails loudly at the call, not at import. one_pole_butter_lowpass / one_pole_filter
# are dead code in the reference (they print to stdout).
fft_freqz = _frequency_domain_helper("fft_freqz", "7-11")
fft_sosfreqz = _frequency_domain_helper("fft_sosfreqz", "14-32")
freqdomain_fir = _frequency_domain_helper("freqdomain_fir", "35-39")
one_pole_butter_lowpass = _frequency_domain_helper("one_pole_butter_lowpass", "169-198")
one_pole_filter = _frequency_domain_helper("one_pole_filter", "201-239")


def sosfilt_via_fsm(sos: torch.Tensor, x: torch.Tensor):
    """Cascade of second-order sections along the last dim of x (reference: signal.py:136-166).

    sos: (bs, n_sections, 6) rows [b0 b1 b2 a0 a1 a2]; bs may be 1 (broadcast). x: (bs, ..., T).
    Differentiable w.r.t. both. Up to 8 sections are one launch per direction (the backward pass is the Gram-matrix kernel for every
    section count: 0.46 ms forward + backward for 8 sections on (256, 2, 131072), profiles/r04/sections_per_call.log); longer cascades are
    applied as successive calls of at most 6 sections."""
    bs, n_sections, n_coeffs = sos.size()
    assert n_coeffs == 6  # must be second order (signal.py:24)
    shape = x.shape
    xx = x.reshape(shape[0], -1, shape[-1])
    if is_f64(x):            # float64 in, float64 arithmetic, as the reference (ops64.py): any number of sections in one call
        return SosFilt64Function.apply(sos, xx).reshape(shape)
    step = 8 if n_sections <= 8 else 6
    for s0 in range(0, n_sections, step):
        xx = _ops.sosfilt(sos[:, s0:s0 + step], xx)         # torch.ops.dasp.sosfilt (csrc/torch_ext), or the ctypes binding
    return xx.reshape(shape)


OCTAVE_BANDS = (31.5, 63, 125, 250, 500, 1000, 2000, 4000, 8000, 16000)


@functools.lru_cache(maxsize=16)
def _octave_band_taps(num_taps: int, sample_rate: float):
    """SciPy window-method design of the 12 filters, exactly the reference's calls (signal.py:60-87); host-side and
    parameter-free, so it is designed once per (num_taps, sample_rate) and cached (the reference redesigns per call)."""
    filts = [scipy.signal.firwin(num_taps, 12, fs=sample_rate)]
    for fc in OCTAVE_BANDS:
        f_min = fc / np.sqrt(2)
        f_max = np.clip(fc * np.sqrt(2), a_min=0, a_max=(sample_rate / 2) * 0.999)
        filts.append(scipy.signal.firwin(num_taps, [f_min, f_max], fs=sample_rate, pass_zero=False))
    filts.append(scipy.signal.firwin(num_taps, 18000, fs=sample_rate, pass_zero=False))
    return np.stack([f.astype("float32")[::-1] for f in filts], 0).copy()   # the reference's torch.flip (a no-op: symmetric)


def octave_band_filterbank(num_taps: int, sample_rate: float):
    """Octave-spaced linear-phase FIR bank, shape (12, 1, num_taps) float32 on the CPU, as the reference
    (dasp_pytorch/signal.py:42-92): lowpass 12 Hz, ten octave bandpasses 31.5 Hz .. 16 kHz, highpass 18 kHz."""
    return torch.from_numpy(_octave_band_taps(int(num_taps), float(sample_rate)).copy()).unsqueeze(1)   # a fresh tensor per call, as the reference
