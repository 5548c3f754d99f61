"""Drop-in for dasp_pytorch.functional on MI355X: same names, argument order and keyword names
(dasp_pytorch/functional.py), every effect computed by hand-written HIP kernels (csrc/)."""
import torch

from .ops import FILTER_TYPES, DistortionFunction, DynamicsFunction, GainFunction, ParametricEQFunction

_PEQ_TYPES = [FILTER_TYPES[t] for t in ("low_shelf", "peaking", "peaking", "peaking", "peaking", "high_shelf")]


def gain(x: torch.Tensor, sample_rate: int, gain_db: torch.Tensor):
    """Apply gain in dB; the same gain is applied to every channel of a batch item
    (reference: dasp_pytorch/functional.py:10-29). gain_db: bs values (any shape that views to (bs, 1, 1))."""
    bs, chs, seq_len = x.size()
    if gain_db.numel() != bs:   # the reference's gain_db.view(bs, 1, 1)
        raise RuntimeError(f"shape '[{bs}, 1, 1]' is invalid for input of size {gain_db.numel()}")
    return GainFunction.apply(x, gain_db)


def distortion(x: torch.Tensor, sample_rate: int, drive_db: torch.Tensor):
    """Soft-clipping distortion tanh(x * 10^(drive_db/20)) (reference: dasp_pytorch/functional.py:65-78).
    As in the reference's drive_db.view(bs, chs, -1), drive_db must hold one value per (batch item,
    channel) row, i.e. bs*chs values (so a (bs,) drive only works for mono input). The reference would
    also accept bs*chs*k values broadcastable against seq_len; that per-sample form is not supported."""
    bs, chs, seq_len = x.size()
    if drive_db.numel() != bs * chs:
        raise RuntimeError(f"shape '[{bs}, {chs}, -1]' is invalid for input of size {drive_db.numel()} "
                           "(dasp_pytorch_amd supports one drive value per (batch, channel) row)")
    return DistortionFunction.apply(x, drive_db)


def parametric_eq(
    x: torch.Tensor,
    sample_rate: float,
    low_shelf_gain_db: torch.Tensor,
    low_shelf_cutoff_freq: torch.Tensor,
    low_shelf_q_factor: torch.Tensor,
    band0_gain_db: torch.Tensor,
    band0_cutoff_freq: torch.Tensor,
    band0_q_factor: torch.Tensor,
    band1_gain_db: torch.Tensor,
    band1_cutoff_freq: torch.Tensor,
    band1_q_factor: torch.Tensor,
    band2_gain_db: torch.Tensor,
    band2_cutoff_freq: torch.Tensor,
    band2_q_factor: torch.Tensor,
    band3_gain_db: torch.Tensor,
    band3_cutoff_freq: torch.Tensor,
    band3_q_factor: torch.Tensor,
    high_shelf_gain_db: torch.Tensor,
    high_shelf_cutoff_freq: torch.Tensor,
    high_shelf_q_factor: torch.Tensor,
):
    """Six-band parametric EQ: low-shelf -> 4 peaking bands -> high-shelf
    (reference: dasp_pytorch/functional.py:118-272). Each control is a tensor with bs (or 1)
    elements; the same filter is applied to every channel of a batch item."""
    bs, chs, seq_len = x.size()
    controls = [
        low_shelf_gain_db, low_shelf_cutoff_freq, low_shelf_q_factor,
        band0_gain_db, band0_cutoff_freq, band0_q_factor,
        band1_gain_db, band1_cutoff_freq, band1_q_factor,
        band2_gain_db, band2_cutoff_freq, band2_q_factor,
        band3_gain_db, band3_cutoff_freq, band3_q_factor,
        high_shelf_gain_db, high_shelf_cutoff_freq, high_shelf_q_factor,
    ]
    n = controls[0].numel()
    if any(c.numel() != n for c in controls) or n not in (1, bs):
        raise RuntimeError(f"parametric_eq controls must each hold {bs} (or 1) values, got {[c.numel() for c in controls]}")
    return ParametricEQFunction.apply(x, float(sample_rate), _PEQ_TYPES, *controls)


def _dynamics(mode, x, sample_rate, threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db, eps, lookahead_samples):
    bs, chs, seq_len = x.size()
    ctls = (threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db)
    for c in ctls:   # the reference's .view(-1, 1, 1) against a (bs, 1, seq_len) side chain: no parameter broadcasting
        if c.numel() != bs:
            raise RuntimeError(f"The size of tensor a ({c.numel()}) must match the size of tensor b ({bs}) at non-singleton dimension 0")
    return DynamicsFunction.apply(x, mode, float(sample_rate), float(eps), int(lookahead_samples), *ctls)


def compressor(
    x: torch.Tensor,
    sample_rate: float,
    threshold_db: torch.Tensor,
    ratio: torch.Tensor,
    attack_ms: torch.Tensor,
    release_ms: torch.Tensor,
    knee_db: torch.Tensor,
    makeup_gain_db: torch.Tensor,
    eps: float = 1e-8,
    lookahead_samples: int = 0,
):
    """Feed-forward dynamic range compressor (reference: dasp_pytorch/functional.py:275-399): summed side
    chain, soft-knee gain computer in dB, one-pole smoothing with the attack time constant (release_ms is
    accepted and ignored, exactly like the reference), optional look-ahead delay of the signal path, make-up
    gain. The smoothing filter is evaluated as an exact recurrence (chunked scan) instead of the reference's
    frequency-sampling FFT filter. Deviation: knee_db == 0 gives finite gradients (the reference's are NaN)."""
    return _dynamics(0, x, sample_rate, threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db, eps, lookahead_samples)


def expander(
    x: torch.Tensor,
    sample_rate: float,
    threshold_db: torch.Tensor,
    ratio: torch.Tensor,
    attack_ms: torch.Tensor,
    release_ms: torch.Tensor,
    knee_db: torch.Tensor,
    makeup_gain_db: torch.Tensor,
    eps: float = 1e-8,
    lookahead_samples: int = 0,
):
    """Downward expander with the compressor's structure and signature. The reference's `expander()` raises
    NotImplementedError (dasp_pytorch/functional.py:402-403), so there is no reference behaviour: below
    threshold - knee/2 the level is mapped to T + (x_db - T) * ratio, with the standard quadratic soft knee
    (Giannoulis, Massberg & Reiss 2012), above the knee the signal is untouched."""
    return _dynamics(1, x, sample_rate, threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db, eps, lookahead_samples)
