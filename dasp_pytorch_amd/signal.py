"""DSP primitives with the reference's names and signatures (dasp_pytorch/signal.py).

`sosfilt_via_fsm` keeps the reference's name for drop-in use, but it is evaluated as an exact
recurrence (chunked parallel scan, csrc/sosfilt.hip) instead of the reference's frequency-sampling
approximation; the two agree to <= 1e-13 in fp64 for stable filters whose impulse response has
decayed within the signal length.
"""
import functools
import math

import numpy as np
import scipy.signal
import torch

from .ops import SosFiltFunction


def biquad(gain_db, cutoff_freq, q_factor, sample_rate, filter_type="peaking"):
    """RBJ-cookbook biquad design; mirrors dasp_pytorch/signal.py:242-306 op for op (tiny (bs,1)
    tensors, left to torch so autograd supplies d(sos)/d(params) when the un-fused path is used).
    Returns b, a with shape (bs, 3), a0-normalised."""
    bs = gain_db.size(0)
    gain_db = gain_db.view(bs, -1)
    cutoff_freq = cutoff_freq.view(bs, -1)
    q_factor = q_factor.view(bs, -1)

    A = 10 ** (gain_db / 40.0)
    w0 = 2 * math.pi * (cutoff_freq / sample_rate)
    alpha = torch.sin(w0) / (2 * q_factor)
    cos_w0 = torch.cos(w0)
    sqrt_A = torch.sqrt(A)

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

    b = torch.stack([b0, b1, b2], dim=1).view(bs, -1)
    a = torch.stack([a0, a1, a2], dim=1).view(bs, -1)
    b = b.type_as(gain_db) / a0
    a = a.type_as(gain_db) / a0
    return b, a


def sosfilt_via_fsm(sos: torch.Tensor, x: torch.Tensor):
    """Cascade of second-order sections along the last dim of x (reference: signal.py:136-166).

    sos: (bs, n_sections, 6) rows [b0 b1 b2 a0 a1 a2]; bs may be 1 (broadcast). x: (bs, ..., T).
    Differentiable w.r.t. both. More than 8 sections are applied as successive <=8-section calls."""
    bs, n_sections, n_coeffs = sos.size()
    assert n_coeffs == 6  # must be second order (signal.py:24)
    shape = x.shape
    xx = x.reshape(shape[0], -1, shape[-1])
    for s0 in range(0, n_sections, 8):
        xx = SosFiltFunction.apply(sos[:, s0:s0 + 8], xx)
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
    return torch.from_numpy(_octave_band_taps(int(num_taps), float(sample_rate))).unsqueeze(1)
