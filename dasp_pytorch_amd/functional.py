"""Drop-in for dasp_pytorch.functional on MI355X: same names, argument order and keyword names
(dasp_pytorch/functional.py), every effect computed by hand-written HIP kernels (csrc/)."""
import torch

from .ops import FILTER_TYPES, DistortionFunction, GainFunction, ParametricEQFunction

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
