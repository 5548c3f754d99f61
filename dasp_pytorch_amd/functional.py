"""Drop-in for dasp_pytorch.functional on MI355X: same names, argument order and keyword names
(dasp_pytorch/functional.py), every effect computed by hand-written HIP kernels (csrc/)."""
import torch

from . import signal as _signal
from . import _mt19937
from . import ops as _ops
from .ops import FILTER_TYPES, BusFunction, DistortionSampleFunction, PannerFunction, WidenerFunction
from .ops import reverb as _ops_reverb
from .ops64 import Dynamics64Function, Elementwise64Function, ParametricEQ64Function, is_f64, require_fp32_ok

_PEQ_TYPES = [FILTER_TYPES[t] for t in ("low_shelf", "peaking", "peaking", "peaking", "peaking", "high_shelf")]


def gain(x: torch.Tensor, sample_rate: int, gain_db: torch.Tensor):
    """Apply gain in dB; the same gain is applied to every channel of a batch item
    (reference: dasp_pytorch/functional.py:10-29). gain_db: bs values (any shape that views to (bs, 1, 1))."""
    bs, chs, seq_len = x.size()
    if gain_db.numel() != bs:   # the reference's gain_db.view(bs, 1, 1)
        raise RuntimeError(f"shape '[{bs}, 1, 1]' is invalid for input of size {gain_db.numel()}")
    if is_f64(x):            # float64 in, float64 arithmetic, as the reference (ops64.py)
        return Elementwise64Function.apply(x, gain_db, 0)
    return _ops.gain(x, gain_db)                      # torch.ops.dasp.gain (csrc/torch_ext), or the ctypes binding


def distortion(x: torch.Tensor, sample_rate: int, drive_db: torch.Tensor):
    """Soft-clipping distortion tanh(x * 10^(drive_db/20)) (reference: dasp_pytorch/functional.py:65-78).
    As in the reference's drive_db.view(bs, chs, -1), drive_db holds one value per (batch item, channel) row, i.e. bs*chs values (so a
    (bs,) drive only works for mono input), or one value per sample, bs*chs*seq_len values; any other count fails as the reference's
    view / broadcast does."""
    bs, chs, seq_len = x.size()
    n = drive_db.numel()
    if n == bs * chs * seq_len and seq_len != 1:
        if is_f64(x):
            require_fp32_ok(x, "distortion with per-sample drive_db")
        return DistortionSampleFunction.apply(x, drive_db)
    if n != bs * chs:
        if bs * chs == 0 or n % (bs * chs) != 0:
            raise RuntimeError(f"shape '[{bs}, {chs}, -1]' is invalid for input of size {n}")
        raise RuntimeError(f"The size of tensor a ({seq_len}) must match the size of tensor b ({n // (bs * chs)}) at non-singleton dimension 2")
    if is_f64(x):
        return Elementwise64Function.apply(x, drive_db, 1)
    return _ops.distortion(x, drive_db)


def stereo_bus(x: torch.Tensor, sample_rate: int, send_db: torch.Tensor):
    """Sum stereo tracks (bs, 2, tracks, seq_len) into one stereo bus (bs, 2, seq_len) with send levels in dB, bs*tracks values
    (reference: dasp_pytorch/functional.py:32-62, send_db.view(bs, 1, tracks, 1)). At most 64 tracks."""
    bs, chs, tracks, seq_len = x.size()
    assert chs == 2, "Input tensor must have shape (bs, 2, tracks, seq_len)"
    if send_db.numel() != bs * tracks:
        raise RuntimeError(f"shape '[{bs}, 1, {tracks}, 1]' is invalid for input of size {send_db.numel()}")
    require_fp32_ok(x, "stereo_bus")
    return BusFunction.apply(x, send_db)


def advanced_distortion(x, sample_rate, input_gain_db, output_gain_db, tone, dc_offset):
    """Not implemented in the reference either (dasp_pytorch/functional.py:81-111)."""
    raise NotImplementedError


def graphic_eq(x: torch.Tensor, sample_rate: float):
    """Not implemented in the reference either (dasp_pytorch/functional.py:114-115)."""
    raise NotImplementedError


def stereo_widener(x: torch.Tensor, sample_rate: float, width: torch.Tensor):
    """Mid/side stereo widener on (bs, 2, seq_len), width with bs values (the reference needs them shaped (bs, 1)): mid * 2 (1 - width), side * 2 width
    (reference: dasp_pytorch/functional.py:580-605)."""
    bs, chs, seq_len = x.size()
    assert chs == 2, "Input tensor must have shape (bs, 2, seq_len)"
    if width.numel() != bs:
        raise RuntimeError(f"The size of tensor a ({bs}) must match the size of tensor b ({width.numel()})")
    require_fp32_ok(x, "stereo_widener")
    return WidenerFunction.apply(x, width)


def stereo_panner(x: torch.Tensor, sample_rate: float, pan: torch.Tensor):
    """Pan mono tracks (bs, num_tracks, seq_len) across the stereo field; pan in [0, 1] with bs*num_tracks values. Returns
    (bs, 2, num_tracks, seq_len), the shape the reference's code produces (its docstring says otherwise;
    dasp_pytorch/functional.py:608-636)."""
    bs, num_tracks, seq_len = x.size()
    if pan.numel() != bs * num_tracks:
        raise RuntimeError(f"shape '[{bs}, 1, {num_tracks}, 1]' is invalid for input of size {pan.numel()}")
    require_fp32_ok(x, "stereo_panner")
    return PannerFunction.apply(x, pan)


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
    if is_f64(x):
        return ParametricEQ64Function.apply(x, float(sample_rate), _PEQ_TYPES, *controls)
    return _ops.parametric_eq(x, sample_rate, _PEQ_TYPES, controls)


def _dynamics(mode, x, sample_rate, threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db, eps, lookahead_samples):
    bs, chs, seq_len = x.size()
    ctls = (threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db)
    for c in ctls:   # the reference's .view(-1, 1, 1) against a (bs, 1, seq_len) side chain: no parameter broadcasting
        if c.numel() != bs:
            raise RuntimeError(f"The size of tensor a ({c.numel()}) must match the size of tensor b ({bs}) at non-singleton dimension 0")
    if is_f64(x):
        return Dynamics64Function.apply(x, mode, float(sample_rate), float(eps), int(lookahead_samples), *ctls)
    return _ops.dynamics(x, mode, sample_rate, eps, lookahead_samples, ctls)


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


_FB_CACHE = {}


def _device_filterbank(num_taps: int, sample_rate: float, device: torch.device):
    """One device-resident copy of the octave filterbank per (taps, sample_rate, device): the taps are constants, so the
    host->device copy and their spectra (ops._filter_spectrum) are paid once, not per call. Not kept when first asked for inside a
    HIP-graph capture (that memory belongs to the graph being captured)."""
    key = (num_taps, sample_rate, device)
    hit = _FB_CACHE.get(key)
    if hit is not None:
        return hit
    bank = _signal.octave_band_filterbank(num_taps, sample_rate).squeeze(1).to(device).contiguous()
    if not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
        if len(_FB_CACHE) >= 8:
            _FB_CACHE.clear()
        _FB_CACHE[key] = bank
    return bank


def noise_shaped_reverberation(
    x: torch.Tensor,
    sample_rate: float,
    band0_gain: torch.Tensor,
    band1_gain: torch.Tensor,
    band2_gain: torch.Tensor,
    band3_gain: torch.Tensor,
    band4_gain: torch.Tensor,
    band5_gain: torch.Tensor,
    band6_gain: torch.Tensor,
    band7_gain: torch.Tensor,
    band8_gain: torch.Tensor,
    band9_gain: torch.Tensor,
    band10_gain: torch.Tensor,
    band11_gain: torch.Tensor,
    band0_decay: torch.Tensor,
    band1_decay: torch.Tensor,
    band2_decay: torch.Tensor,
    band3_decay: torch.Tensor,
    band4_decay: torch.Tensor,
    band5_decay: torch.Tensor,
    band6_decay: torch.Tensor,
    band7_decay: torch.Tensor,
    band8_decay: torch.Tensor,
    band9_decay: torch.Tensor,
    band10_decay: torch.Tensor,
    band11_decay: torch.Tensor,
    mix: torch.Tensor,
    num_samples: int = 65536,
    num_bandpass_taps: int = 1023,
    noise: torch.Tensor = None,
    device_noise: bool = False,
    noise_seed: int = None,
    noise_seed_offset: torch.Tensor = None,
):
    """Artificial reverberation from frequency-band noise shaping (reference: dasp_pytorch/functional.py:406-577).
    Mono input is duplicated to stereo and the output always has 2 channels, as in the reference.

    White noise: by default it is what the reference draws -- torch.randn(bs*2, 12, num_samples + num_bandpass_taps - 1) from the
    global *CPU* generator (functional.py:548) -- so the same torch.manual_seed gives the same impulse responses as the reference,
    and the CPU generator is left in the state that call leaves it in; the values are computed on x's device from the generator's
    state (csrc/mtrand.hip, _mt19937.py: the twister run in parallel by jump-ahead, torch's float and Box-Muller layout), not drawn
    on the host and copied (0.56 s + 0.8 GB at (128,2,262144)). `device_noise=True` generates it on x's device
    instead, inside the filter-bank kernels (a counter-based stream, csrc/reverb.hip: the noise tensor - 0.8 GB at the default sizes
    and 128 items - never exists, forward and backward recompute it); its 63-bit seed is `noise_seed` (giving a seed selects this
    mode by itself), or, when that is None, one draw
    from torch's global CPU generator per call - so torch.manual_seed makes it reproducible and successive calls differ, as with the
    reference. Inside a HIP-graph capture that draw happens once, at capture time; `noise_seed_offset`, a 1-element int64 tensor
    on x's device, is added to the seed when the kernels run - bump it once per replay (e.g. `offset.add_(1)` at the end of the
    captured step) and every replay draws new noise. `noise=` supplies the noise tensor explicitly. The keywords are additions; Processor.process_normalized passes only the named parameters above."""
    assert num_bandpass_taps % 2 == 1, "num_bandpass_taps must be odd"
    bs, chs, seq_len = x.size()
    assert chs <= 2, "only mono/stereo signals are supported"
    require_fp32_ok(x, "noise_shaped_reverberation")
    for c in (band0_gain, band1_gain, band2_gain, band3_gain, band4_gain, band5_gain, band6_gain, band7_gain, band8_gain, band9_gain,
              band10_gain, band11_gain, band0_decay, band1_decay, band2_decay, band3_decay, band4_decay, band5_decay, band6_decay,
              band7_decay, band8_decay, band9_decay, band10_decay, band11_decay, mix):
        if c.numel() != bs:   # the reference's torch.stack(...).view(bs, 12) / mix.view(bs, 1, 1) (functional.py:498-544): no broadcasting
            raise RuntimeError(f"shape '[{bs}, 12]' is invalid for input of size {12 * c.numel()}")
    # (chs == 1: the reference copies mono to stereo, functional.py:493-495; the kernels read the one row for both output channels,
    # ops.ReverbFunction)
    gains = (band0_gain, band1_gain, band2_gain, band3_gain, band4_gain, band5_gain, band6_gain, band7_gain, band8_gain, band9_gain, band10_gain, band11_gain)
    decays = (band0_decay, band1_decay, band2_decay, band3_decay, band4_decay, band5_decay, band6_decay, band7_decay, band8_decay, band9_decay, band10_decay,
              band11_decay)
    if _ops.noise_shaped_reverb_ok(x, gains, decays, mix, noise):
        # the 25 tensors as they are: torch.ops.dasp.noise_shaped_reverb stacks them and hands their gradients back in C++ (csrc/torch_ext)
        filters, noise, seed, offset, pending = _reverb_noise(x, sample_rate, num_samples, num_bandpass_taps, noise, device_noise, noise_seed, noise_seed_offset)
        try:
            return _ops.noise_shaped_reverb(x, noise, filters, gains, decays, mix, int(num_samples), seed, offset, 0.0)
        finally:
            pending.finish()
    return _reverb_from_matrices(x, sample_rate, _StackColumns.apply(*gains), _StackColumns.apply(*decays), mix.view(bs), num_samples, num_bandpass_taps, noise,
                                 device_noise, noise_seed, noise_seed_offset)


class _StackColumns(torch.autograd.Function):
    """torch.stack([c.view(bs) for c in cols], dim=1) with a backward that hands every column its gradient as a contiguous row of one
    transposed matrix: autograd's own stack backward returns 12 strided column views, each of which AccumulateGrad then copies with a
    kernel of its own (24 launches per reverb step for the 2 x 12 band controls; here: one transpose per matrix)."""

    @staticmethod
    def forward(ctx, *cols):
        ctx.shapes = [c.shape for c in cols]
        return torch.stack([c.reshape(-1) for c in cols], dim=1)

    @staticmethod
    def backward(ctx, g):
        rows = g.t().contiguous().unbind(0)
        return tuple(r.reshape(shape) if need else None for r, shape, need in zip(rows, ctx.shapes, ctx.needs_input_grad))


def _reverb_noise(x, sample_rate, num_samples, num_bandpass_taps, noise, device_noise, noise_seed, noise_seed_offset):
    """The filter bank on x's device and the white noise of one call, as the reference draws it (functional.py:548) or as the keywords of
    noise_shaped_reverberation ask for it: (filters, noise tensor or None, seed or None, seed offset or None, pending). `pending.finish()`
    installs the CPU generator's state after a default draw (its read-back rides behind the generating kernels): call it once the
    forward kernels are queued, before returning to the caller."""
    filters = _device_filterbank(int(num_bandpass_taps), float(sample_rate), x.device)
    seed, pending = None, _mt19937._NothingPending()
    if noise_seed_offset is not None and (noise is not None or not (device_noise or noise_seed is not None)):
        raise ValueError("noise_seed_offset only applies to the generated noise (device_noise=True or noise_seed=...)")
    if noise is None:
        if device_noise or noise_seed is not None:
            # one 63-bit draw from the global CPU generator (a host-side scalar: no device work, no sync) unless the caller fixed the seed
            seed = int(noise_seed) if noise_seed is not None else int(torch.empty((), dtype=torch.int64).random_().item())
        else:
            # the reference's draw (functional.py:548), made where it is used: the values and the CPU generator's state afterwards are
            # those of torch.randn(bs*2, 12, ...) on the global CPU generator, computed by csrc/mtrand.hip from that generator's state
            noise, pending = _mt19937.randn_cpu_stream(x.shape[0] * 2, 12, num_samples + num_bandpass_taps - 1, device=x.device, defer=True)
    return filters, noise, seed, (noise_seed_offset if seed is not None else None), pending


def _reverb_from_matrices(x, sample_rate, band_gains, band_decays, mix, num_samples=65536, num_bandpass_taps=1023, noise=None, device_noise=False,
                          noise_seed=None, noise_seed_offset=None, decay_bound=0.0):
    """noise_shaped_reverberation on the band gains / decays as (bs, 12) matrices and mix (bs): what NoiseShapedReverb.process_normalized has
    as slices of its de-normalised (bs, 25) tensor. x: (bs, 1 or 2, seq_len).
    decay_bound > 0: the caller vouches that no band decay exceeds it (a validated parameter range) - lets the filter bank skip a launch."""
    filters, noise, seed, offset, pending = _reverb_noise(x, sample_rate, num_samples, num_bandpass_taps, noise, device_noise, noise_seed, noise_seed_offset)
    try:
        return _ops_reverb(x, noise, filters, band_gains.to(x.device), band_decays.to(x.device), mix.to(x.device), int(num_samples), seed, offset, float(decay_bound))
    finally:
        pending.finish()


def _dynamics_from_matrix(mode, x, sample_rate, controls, eps=1e-8, lookahead_samples=0):
    """compressor / expander on the (bs, 6) matrix of controls (columns in the functions' argument order)."""
    from .ops import DynamicsMatrixFunction
    return DynamicsMatrixFunction.apply(x, mode, float(sample_rate), float(eps), int(lookahead_samples), controls)
