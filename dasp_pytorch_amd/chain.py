"""The effect chain of the reference's style-transfer model with its last stage folded away.

The reference runs `equalizer -> compressor -> reverb -> gain`, four passes over (bs, chs, seq_len) forward and four backward
(examples/style_transfer.py:150-154). The noise-shaped reverb is linear and time-invariant per batch item, so a per-item gain
commutes with it:  gain * reverb(c) = reverb(gain * c);  and a gain applied to the compressor's output is what its make-up gain
already does:  gain_db just adds to makeup_gain_db. The chain below therefore runs three kernels' worth of passes, not four, with
bit-for-bit the same mathematics (up to fp32 rounding of one multiply) and the gradient of gain_db falling out of the make-up gain's.
"""
import os

import torch

from . import functional as _functional
from . import modules as _modules


class StyleTransferChain:
    """EQ -> compressor -> reverb -> gain on normalised parameters, the gain folded into the compressor's make-up gain.

    Same constructor idea and `process_normalized` contract as the four `Processor`s it replaces (`dasp_pytorch/modules.py`):
    `process_normalized(x, eq_params (bs, 18), comp_params (bs, 6), reverb_params (bs, 25), gain_params (bs, 1))`, every entry in
    [0, 1]; mono or stereo `x`, stereo output (the reverb's)."""

    def __init__(self, sample_rate, **reverb_kwargs):
        self.sample_rate = sample_rate
        self.equalizer = _modules.ParametricEQ(sample_rate)
        self.compressor = _modules.Compressor(sample_rate)
        self.reverb = _modules.NoiseShapedReverb(sample_rate, **reverb_kwargs)
        self.gain = _modules.Gain(sample_rate)
        self.num_params = (self.equalizer.num_params, self.compressor.num_params, self.reverb.num_params, self.gain.num_params)

    def process_normalized(self, x: torch.Tensor, eq_params, comp_params, reverb_params, gain_params):
        for proc, p in ((self.equalizer, eq_params), (self.compressor, comp_params), (self.reverb, reverb_params), (self.gain, gain_params)):
            if p.shape[1] != proc.num_params:
                raise ValueError(f"Parameter tensor has {p.shape[1]} parameters, but processor has {proc.num_params} parameters.")
        # one [0, 1] check for all 50 parameters - one reduction, one read-back, before any kernel of the chain is queued
        procs = (self.equalizer, self.compressor, self.reverb, self.gain)
        if all(p.validate_range for p in procs) and not (x.is_cuda and torch.cuda.is_current_stream_capturing()):
            every = (eq_params, comp_params, reverb_params, gain_params)
            if all(t.dim() == 2 and t.shape[0] == every[0].shape[0] and t.dtype == every[0].dtype and t.device == every[0].device for t in every):
                _modules.check_unit_range(torch.cat([t.detach() for t in every], dim=1), [n for p in procs for n in p.param_ranges])
                with _modules.already_validated():
                    return self._run(x, eq_params, comp_params, reverb_params, gain_params)
        return self._run(x, eq_params, comp_params, reverb_params, gain_params)

    def _tables(self):
        """lo / span of the compressor's 6, the reverb's 25 and the gain's 1 parameter as the two float[32] arrays dasp_chain_controls takes."""
        ranges = list(self.compressor.param_ranges.values()) + list(self.reverb.param_ranges.values()) + list(self.gain.param_ranges.values())
        key = tuple(ranges)
        if getattr(self, "_tab_key", None) != key:
            import ctypes
            self._tab = ((ctypes.c_float * 32)(*[float(r[0]) for r in ranges]), (ctypes.c_float * 32)(*[float(r[1]) - float(r[0]) for r in ranges]))
            self._tab_key = key
        return self._tab

    def _run(self, x, eq_params, comp_params, reverb_params, gain_params):
        m = _modules
        every = (comp_params, reverb_params, gain_params)
        # only the EQ broadcasts a parameter batch of 1 (functional.py:208-220); compressor, reverb and gain raise in the reference
        # (their .view(bs, ...) / side-chain broadcast), and the kernels read one row of controls per batch item
        for name, t in (("compressor", comp_params), ("reverb", reverb_params), ("gain", gain_params)):
            if t.dim() != 2 or t.shape[0] != x.shape[0]:
                raise RuntimeError(f"The size of tensor a ({t.shape[0] if t.dim() else 1}) must match the size of tensor b ({x.shape[0]}) at "
                                   f"non-singleton dimension 0 ({name} parameters: one row per batch item, got {tuple(t.shape)})")
        fused = (os.environ.get("DASP_CHAIN_FUSED_CONTROLS", "1") != "0"        # developer A/B: the torch-op de-normalisation below
                 and x.is_cuda and x.dtype is torch.float32 and x.dim() == 3 and x.shape[1] <= 2
                 and all(t.is_cuda and t.dtype is torch.float32 and t.dim() == 2 and t.shape[0] == x.shape[0] for t in every)
                 and comp_params.shape[1] == 6 and reverb_params.shape[1] == 25 and gain_params.shape[1] == 1
                 and list(self.compressor.param_ranges) == m._DYN_NAMES and list(self.reverb.param_ranges) == m._REV_NAMES
                 and self.compressor.process_fn is _functional.compressor and self.reverb.process_fn is self.reverb._rev_fn
                 and self.gain.process_fn is _functional.gain)
        if fused:
            # every control of the three stages behind the EQ from one launch (and one back): ops.ChainControlsFunction
            from .ops import ChainControlsFunction, DynamicsCtlFunction
            self.gain._check_range(gain_params)
            self.compressor._check_range(comp_params)
            self.reverb._check_range(reverb_params)
            lo, span = self._tables()
            ctl, gains, decays, mix = ChainControlsFunction.apply(comp_params, reverb_params, gain_params, lo, span)
            eq = self.equalizer
            no_grad = not (torch.is_grad_enabled() and (x.requires_grad or eq_params.requires_grad or ctl.requires_grad))
            if (no_grad and os.environ.get("DASP_CHAIN_FUSED_FORWARD", "1") != "0" and eq.process_fn is _functional.parametric_eq
                    and list(eq.param_ranges) == m._EQ_NAMES and eq_params.dim() == 2 and eq_params.shape[1] == 18
                    and eq_params.shape[0] in (1, x.shape[0]) and eq_params.is_cuda and eq_params.is_floating_point()):
                # forward only (the reference's target synthesis, examples/style_transfer.py:293-299): EQ and compressor as ONE pass over x
                # (csrc/chainfwd.hip) - the EQ's output never goes to memory
                from .ops import chain_eq_compressor_forward
                eq._check_range(eq_params)
                elo = [float(r[0]) for r in eq.param_ranges.values()]
                espan = [float(r[1]) - float(r[0]) for r in eq.param_ranges.values()]
                y = chain_eq_compressor_forward(x, eq_params, _functional._PEQ_TYPES, elo, espan, float(self.sample_rate), ctl)
            else:
                y = eq.process_normalized(x, eq_params)                     # fused de-normalise + design; no gradient for x: the no-gx kernel
                y = DynamicsCtlFunction.apply(y, 0, float(self.sample_rate), 1e-8, 0, ctl)
            # (mono: the reverb kernels read the one row for both output channels - no duplicated copy, functional.py:493-495)
            return _functional._reverb_from_matrices(y, self.sample_rate, gains, decays, mix, decay_bound=self.reverb._decay_bound(), **self.reverb._rev_kwargs)
        self.gain._check_range(gain_params)
        self.compressor._check_range(comp_params)
        lo, span = self.gain._affine(gain_params)
        gain_db = gain_params * span + lo                                   # (bs, 1), differentiable
        clo, cspan = self.compressor._affine(comp_params)
        comp = comp_params * cspan + clo                                    # denormalised compressor controls, columns in param_ranges order
        y = self.equalizer.process_normalized(x, eq_params)                 # fused de-normalise + design; no gradient for x: the no-gx kernel
        names = list(self.compressor.param_ranges)
        if y.is_cuda and y.dtype is torch.float32 and names == _modules._DYN_NAMES:
            folded = torch.cat([comp[:, :5], comp[:, 5:] + gain_db], dim=1)                            # the fold: make-up gain += gain
            y = _functional._dynamics_from_matrix(0, y, self.sample_rate, folded)
        else:
            kwargs = {n: comp[:, i] for i, n in enumerate(names)}
            kwargs["makeup_gain_db"] = kwargs["makeup_gain_db"] + gain_db[:, 0]
            y = self.compressor.process_fn(y, self.sample_rate, **kwargs)
        return self.reverb.process_normalized(y, reverb_params)
