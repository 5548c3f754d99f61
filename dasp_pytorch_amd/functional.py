"""Drop-in for dasp_pytorch.functional on MI355X: same names, argument order and keyword names
(dasp_pytorch/functional.py), every effect computed by hand-written HIP kernels (csrc/)."""
import torch

from .ops import FILTER_TYPES, ParametricEQFunction

_PEQ_TYPES = [FILTER_TYPES[t] for t in ("low_shelf", "peaking", "peaking", "peaking", "peaking", "high_shelf")]


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
