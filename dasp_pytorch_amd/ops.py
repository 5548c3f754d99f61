"""The op layer of the package: every effect of functional.py / signal.py / modules.py / chain.py goes through one of these functions.

float32 ROCm tensors take torch.ops.dasp.* - the PyTorch extension (csrc/torch_ext/dasp_torch_ops.cpp -> csrc/libdasp_torch.so): schemas
registered with TORCH_LIBRARY, forward + hand-derived adjoint as C++ torch::autograd::Function over the C ABI, fake kernels for
torch.compile, opcheck-clean. Everything else - float64, the ops the extension does not register, config.plan.torch_ops = False, a tree
without libdasp_torch.so - takes the ctypes autograd.Functions of _ctypes_ops.py (same kernels, same C entry points; what changes is
the host side: ctypes marshalling and Python in the backward pass), which also own the dtype / device error messages.
"""
import ctypes

import torch

from . import _lib, config
from ._ctypes_ops import (FILTER_TYPES, BiquadFunction, BusFunction, ChainControlsFunction, DistortionFunction, DistortionSampleFunction,   # noqa: F401
                          DynamicsCtlFunction, DynamicsFunction, DynamicsMatrixFunction, GainFunction, PannerFunction, ParametricEQFunction,
                          ParametricEQNormFunction, ReverbFunction, SosFiltFunction, WidenerFunction, _dyn_counters, _filter_spectrum, _seed_offset,
                          _SosWork, chain_eq_compressor_forward, reverb_noise)
from ._lib import check


def _torch_ops_ok(*tensors):
    from . import _torch_ops
    # (under torch.compile the binding is not a choice: only the registered ops can be traced - and the environment / timer look-ups of
    # enabled() would be graph breaks)
    on = _torch_ops._state["loaded"] is True if torch.compiler.is_compiling() else _torch_ops.enabled()
    return on and all(t is None or (t.is_cuda and t.dtype is torch.float32) for t in tensors)


def _same_device_tensors(x, tensors):
    return all(isinstance(t, torch.Tensor) and t.device == x.device for t in tensors)


def parametric_eq(x, sample_rate, types, controls):
    """functional.parametric_eq's core: ParametricEQFunction, or torch.ops.dasp.parametric_eq (float32 audio and every control on its device)."""
    if _torch_ops_ok(x) and x.dim() == 3 and _same_device_tensors(x, controls) and len(types) in (2, 4, 6, 8):
        return torch.ops.dasp.parametric_eq(x, float(sample_rate), list(controls), list(types))
    return ParametricEQFunction.apply(x, float(sample_rate), types, *controls)


def dynamics(x, mode, sample_rate, eps, lookahead, controls):
    """functional.compressor / expander's core on the six control tensors: DynamicsFunction, or torch.ops.dasp.dynamics."""
    if _torch_ops_ok(x) and x.dim() == 3 and _same_device_tensors(x, controls):
        return torch.ops.dasp.dynamics(x, float(sample_rate), *controls, float(eps), int(lookahead), int(mode))
    return DynamicsFunction.apply(x, mode, float(sample_rate), float(eps), int(lookahead), *controls)


def gain(x, gain_db):
    """GainFunction, or torch.ops.dasp.gain."""
    if _torch_ops_ok(x) and x.dim() == 3 and _same_device_tensors(x, (gain_db,)):
        return torch.ops.dasp.gain(x, gain_db)
    return GainFunction.apply(x, gain_db)


def distortion(x, drive_db):
    """DistortionFunction (one drive value per (item, channel) row), or torch.ops.dasp.distortion."""
    if _torch_ops_ok(x) and x.dim() == 3 and _same_device_tensors(x, (drive_db,)):
        return torch.ops.dasp.distortion(x, drive_db)
    return DistortionFunction.apply(x, drive_db)


def sosfilt(sos, x):
    """One cascade call of at most 8 sections on x (bs, chs, seq_len): SosFiltFunction, or torch.ops.dasp.sosfilt."""
    if _torch_ops_ok(x) and x.dim() == 3 and _same_device_tensors(x, (sos,)) and sos.dim() == 3 and 1 <= sos.shape[1] <= 8 and sos.is_floating_point():
        return torch.ops.dasp.sosfilt(sos, x)
    return SosFiltFunction.apply(sos, x)


def parametric_eq_norm(x, pn, sample_rate, types, lo, span, range_flag=None):
    """ParametricEQNormFunction, or torch.ops.dasp.parametric_eq_norm. range_flag: one int32 device word into which the design kernel ORs
    bit i when column i of pn holds a value outside [0, 1] (modules._FlagRangeCheck reads it back one call later)."""
    if _torch_ops_ok(x) and x.dim() == 3 and pn.is_cuda and pn.is_floating_point() and pn.device == x.device:
        return torch.ops.dasp.parametric_eq_norm(x, pn, float(sample_rate), list(types), list(lo), list(span), range_flag)
    return ParametricEQNormFunction.apply(x, pn, float(sample_rate), types, lo, span, range_flag)


def dynamics_ctl(x, mode, sample_rate, eps, lookahead, ctl):
    """DynamicsCtlFunction, or torch.ops.dasp.dynamics_ctl."""
    if _torch_ops_ok(x, ctl) and x.dim() == 3 and ctl.device == x.device:
        return torch.ops.dasp.dynamics_ctl(x, ctl, int(mode), float(sample_rate), float(eps), int(lookahead))
    return DynamicsCtlFunction.apply(x, mode, sample_rate, eps, lookahead, ctl)


# one workgroup per item: the forward pair alone (256,2,131072) 0.272 -> 0.207 ms, (256,1,131072) 0.144 -> 0.133, (128,2,131072) 0.176 -> 0.183
# (profiles/r06/chain_fwd_saving_ab.log); inside the whole chain step, whose reverb is 3 ms at these sizes: (256,2) 3.666 -> 3.651,
# (192,2) 2.868 -> 2.851, (256,1) 3.378 -> 3.389, (128,2) 1.980 -> 1.999 (chain_step_ab.log) - taken from 384 rows on
EQ_DYN_FUSED_MIN_ROWS = 384


def eq_dynamics_norm_ok(x, pn, ctl):
    """True when EQ -> compressor should run as ONE forward pass that saves for both backward passes (torch.ops.dasp.eq_dyn_norm,
    csrc/chainfwd.hip dasp_chain_forward_saving): the torch extension's binding, six sections, one or two channels, and enough items for one
    workgroup per item to fill the device (config.plan.chain_fused_grad: None = from EQ_DYN_FUSED_MIN_ROWS rows = items x channels on,
    True = always, False = never)."""
    mode = config.plan.chain_fused_grad
    if mode is False or not (_torch_ops_ok(x, pn, ctl) and x.dim() == 3 and x.shape[1] <= 2 and x.numel() and pn.dim() == 2 and pn.shape[1] == 18
                             and pn.shape[0] in (1, x.shape[0]) and ctl.dim() == 2 and ctl.shape == (x.shape[0], 5) and pn.device == x.device == ctl.device):
        return False
    return bool(mode) or x.shape[0] * x.shape[1] >= EQ_DYN_FUSED_MIN_ROWS


def eq_dynamics_norm(x, pn, types, lo, span, sample_rate, ctl, mode=0, eps=1e-8, range_flag=None):
    """compressor(parametric_eq(x)) from the normalised (bs, 18) EQ tensor and the (bs, 5) compressor rows, with gradients for x, pn and
    ctl: one forward pass (15 B per channel-sample instead of 19), the two existing backward passes. Call when eq_dynamics_norm_ok."""
    return torch.ops.dasp.eq_dyn_norm(x, pn, float(sample_rate), [int(t) for t in types], [float(v) for v in lo], [float(v) for v in span], ctl,
                                      int(mode), float(eps), range_flag)


def chain_controls(comp_pn, reverb_pn, gain_pn, lo, span, range_flag=None):
    """ChainControlsFunction (lo, span: ctypes float[32]), or torch.ops.dasp.chain_controls. range_flag: as for parametric_eq_norm, bit i for
    column i of the 32 (compressor 0-5, reverb 6-30, gain 31)."""
    if _torch_ops_ok(comp_pn, reverb_pn, gain_pn) and comp_pn.device == reverb_pn.device == gain_pn.device:
        return torch.ops.dasp.chain_controls(comp_pn, reverb_pn, gain_pn, list(lo), list(span), range_flag)
    return ChainControlsFunction.apply(comp_pn, reverb_pn, gain_pn, lo, span, range_flag)


def _reverb_fspec(x, filters, L_ir):
    nb, taps = filters.shape
    with torch.cuda.device(x.device):
        sizes = (ctypes.c_long * 14)()
        check(_lib.lib().dasp_reverb_sizes(x.shape[0], x.shape[2], int(L_ir), taps, nb, sizes), "dasp_reverb_sizes")
        return _filter_spectrum(filters, nb, taps, sizes[4], x.device)


def _signed64(seed):
    s = 0 if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF
    return s - (1 << 64) if s >= (1 << 63) else s                   # the op's `int` is a signed 64-bit word: same bits


def reverb(x, noise, filters, gains, decays, mix, L_ir, seed=None, seed_offset=None, decay_bound=0.0):
    """ReverbFunction, or torch.ops.dasp.reverb (the filter spectra stay cached on the Python side and go in as a tensor)."""
    if (_torch_ops_ok(x, noise, gains, decays, mix) and x.dim() == 3 and x.shape[1] in (1, 2) and x.numel() and filters.is_cuda
            and all(t.device == x.device for t in (filters, gains, decays, mix)) and (noise is None or noise.device == x.device)
            and not filters.requires_grad and (noise is None or not noise.requires_grad)):
        if noise is None and seed is None:
            raise ValueError("reverb: either a noise tensor or a seed")
        nb, taps = filters.shape
        return torch.ops.dasp.reverb(x, noise, _reverb_fspec(x, filters, L_ir), gains, decays, mix, int(L_ir), int(taps), int(nb), _signed64(seed),
                                     _seed_offset(seed_offset, x.device) if noise is None else None, float(decay_bound))
    return ReverbFunction.apply(x, noise, filters, gains, decays, mix, L_ir, seed, seed_offset, decay_bound)


def noise_shaped_reverb_ok(x, band_gains, band_decays, mix, noise):
    """True when torch.ops.dasp.noise_shaped_reverb takes these tensors as they are (float32 audio on a ROCm device, the 25 floating-point
    controls and an explicit noise tensor, if any, on that device)."""
    ctls = list(band_gains) + list(band_decays) + [mix]
    return (_torch_ops_ok(x, noise) and x.dim() == 3 and x.shape[1] in (1, 2) and x.numel() > 0 and _same_device_tensors(x, ctls)
            and all(t.is_floating_point() for t in ctls) and (noise is None or (noise.device == x.device and not noise.requires_grad)))


def noise_shaped_reverb(x, noise, filters, band_gains, band_decays, mix, L_ir, seed=None, seed_offset=None, decay_bound=0.0):
    """functional.noise_shaped_reverberation's core on its 12 + 12 + 1 control tensors (each with bs values): torch.ops.dasp.noise_shaped_reverb
    (the stacks and their backward in C++). Call it when noise_shaped_reverb_ok says so."""
    if noise is None and seed is None:
        raise ValueError("reverb: either a noise tensor or a seed")
    taps = filters.shape[1]
    return torch.ops.dasp.noise_shaped_reverb(x, list(band_gains), list(band_decays), mix, noise, _reverb_fspec(x, filters, L_ir), int(L_ir), int(taps),
                                              _signed64(seed), _seed_offset(seed_offset, x.device) if noise is None else None, float(decay_bound))
