"""The effect chain of the reference's style-transfer model with its last stage folded away.

The reference runs `equalizer -> compressor -> reverb -> gain`, four passes over (bs, chs, seq_len) forward and four backward
(examples/style_transfer.py:150-154). The noise-shaped reverb is linear and time-invariant per batch item, so a per-item gain
commutes with it:  gain * reverb(c) = reverb(gain * c);  and a gain applied to the compressor's output is what its make-up gain
already does:  gain_db just adds to makeup_gain_db. The chain below therefore runs three kernels' worth of passes, not four, with
bit-for-bit the same mathematics (up to fp32 rounding of one multiply) and the gradient of gain_db falling out of the make-up gain's.
"""

import torch

from . import config

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
        self._flag_check = _modules._FlagRangeCheck(2)       # word 0: the EQ's 18 columns, word 1: compressor 0-5, reverb 6-30, gain 31

    def process_normalized(self, x: torch.Tensor, eq_params, comp_params, reverb_params, gain_params):
        for proc, p in ((self.equalizer, eq_params), (self.compressor, comp_params), (self.reverb, reverb_params), (self.gain, gain_params)):
            if p.shape[1] != proc.num_params:
                raise ValueError(f"Parameter tensor has {p.shape[1]} parameters, but processor has {proc.num_params} parameters.")
        # one [0, 1] check for all 50 parameters - one reduction, one read-back, before any kernel of the chain is queued
        procs = (self.equalizer, self.compressor, self.reverb, self.gain)
        if all(p.validate_range for p in procs) and not (x.is_cuda and torch.cuda.is_current_stream_capturing()):
            every = (eq_params, comp_params, reverb_params, gain_params)
            if all(t.dim() == 2 and t.shape[0] == every[0].shape[0] and t.dtype == every[0].dtype and t.device == every[0].device for t in every):
                names = [n for p in procs for n in p.param_ranges]
                deferred = all(p.validate_range == "deferred" for p in procs)
                if deferred and self._fused_ok(x, comp_params, reverb_params, gain_params) and self.equalizer._fused_ok(x, eq_params):
                    # no host wait and no launch: the EQ's design kernel and the chain's control kernel flag the columns outside [0, 1]
                    # themselves, two device words that are read back one call late (modules._FlagRangeCheck)
                    fc = self._flag_check
                    fc.begin()
                    with _modules.already_validated(enforced=False), torch.cuda.device(x.device):
                        y = self._run(x, eq_params, comp_params, reverb_params, gain_params, flags=fc.words(x.device))
                        fc.end([list(self.equalizer.param_ranges), names[self.equalizer.num_params:]])
                    return y
                if deferred:      # no host wait: the previous call's numbers are looked at (modules._DeferredRangeCheck)
                    self.equalizer._deferred().submit(torch.cat([t.detach() for t in every], dim=1), names)
                else:
                    _modules.check_unit_range(torch.cat([t.detach() for t in every], dim=1), names)
                with _modules.already_validated(enforced=not deferred):
                    return self._run(x, eq_params, comp_params, reverb_params, gain_params)
        return self._run(x, eq_params, comp_params, reverb_params, gain_params)

    def flush_range_check(self):
        """validate_range = "deferred": raise now if the last call's parameters were outside [0, 1]."""
        self._flag_check.flush()
        for p in (self.equalizer, self.compressor, self.reverb, self.gain):
            p.flush_range_check()

    def _fused_ok(self, x, comp_params, reverb_params, gain_params):
        """The three stages behind the EQ take their controls from one launch (ops.chain_controls): float32 tensors on x's device, the
        reference's parameter names, the stock process functions."""
        m = _modules
        every = (comp_params, reverb_params, gain_params)
        return (config.plan.chain_fused_controls        # developer A/B: the torch-op de-normalisation below
                and x.is_cuda and x.dtype is torch.float32 and x.dim() == 3 and x.shape[1] <= 2
                and all(t.is_cuda and t.dtype is torch.float32 and t.dim() == 2 and t.shape[0] == x.shape[0] and t.device == x.device for t in every)
                and comp_params.shape[1] == 6 and reverb_params.shape[1] == 25 and gain_params.shape[1] == 1
                and list(self.compressor.param_ranges) == m._DYN_NAMES and list(self.reverb.param_ranges) == m._REV_NAMES
                and self.compressor.process_fn is _functional.compressor and self.reverb.process_fn is self.reverb._rev_fn
                and self.gain.process_fn is _functional.gain)

    def _tables(self):
        """lo / span of the compressor's 6, the reverb's 25 and the gain's 1 parameter as the two float[32] arrays dasp_chain_controls takes."""
        ranges = list(self.compressor.param_ranges.values()) + list(self.reverb.param_ranges.values()) + list(self.gain.param_ranges.values())
        key = tuple(ranges)
        if getattr(self, "_tab_key", None) != key:
            import ctypes
            self._tab = ((ctypes.c_float * 32)(*[float(r[0]) for r in ranges]), (ctypes.c_float * 32)(*[float(r[1]) - float(r[0]) for r in ranges]))
            self._tab_key = key
        return self._tab

    def _run(self, x, eq_params, comp_params, reverb_params, gain_params, flags=None):
        m = _modules
        # only the EQ broadcasts a parameter batch of 1 (functional.py:208-220); compressor, reverb and gain raise in the reference
        # (their .view(bs, ...) / side-chain broadcast), and the kernels read one row of controls per batch item
        for name, t in (("compressor", comp_params), ("reverb", reverb_params), ("gain", gain_params)):
            if t.dim() != 2 or t.shape[0] != x.shape[0]:
                raise RuntimeError(f"The size of tensor a ({t.shape[0] if t.dim() else 1}) must match the size of tensor b ({x.shape[0]}) at "
                                   f"non-singleton dimension 0 ({name} parameters: one row per batch item, got {tuple(t.shape)})")
        fused = self._fused_ok(x, comp_params, reverb_params, gain_params)
        if fused:
            # every control of the three stages behind the EQ from one launch (and one back): ops.ChainControlsFunction
            from . import ops as _ops
            self.gain._check_range(gain_params)
            self.compressor._check_range(comp_params)
            self.reverb._check_range(reverb_params)
            lo, span = self._tables()
            ctl, gains, decays, mix = _ops.chain_controls(comp_params, reverb_params, gain_params, lo, span, None if flags is None else flags[1:2])
            eq = self.equalizer
            no_grad = not (torch.is_grad_enabled() and (x.requires_grad or eq_params.requires_grad or ctl.requires_grad))
            if (no_grad and config.plan.chain_fused_forward and eq.process_fn is _functional.parametric_eq
                    and list(eq.param_ranges) == m._EQ_NAMES and eq_params.dim() == 2 and eq_params.shape[1] == 18
                    and eq_params.shape[0] in (1, x.shape[0]) and eq_params.is_cuda and eq_params.is_floating_point()):
                # forward only (the reference's target synthesis, examples/style_transfer.py:293-299): EQ and compressor as ONE pass over x
                # (csrc/chainfwd.hip) - the EQ's output never goes to memory
                from .ops import chain_eq_compressor_forward
                eq._check_range(eq_params)
                elo = [float(r[0]) for r in eq.param_ranges.values()]
                espan = [float(r[1]) - float(r[0]) for r in eq.param_ranges.values()]
                y = chain_eq_compressor_forward(x, eq_params, _functional._PEQ_TYPES, elo, espan, float(self.sample_rate), ctl,
                                                range_flag=None if flags is None else flags[0:1])
            elif (not no_grad and eq.process_fn is _functional.parametric_eq and list(eq.param_ranges) == m._EQ_NAMES
                  and _ops.eq_dynamics_norm_ok(x, eq_params, ctl)):
                # the pass with gradients at 384 rows (192 stereo items) and more: EQ and compressor as ONE forward pass that writes what the two backward passes
                # read (the EQ's output, its chunk states, the compressor's tile carries) - 15 B per channel-sample instead of 19
                eq._check_range(eq_params)
                elo = [float(r[0]) for r in eq.param_ranges.values()]
                espan = [float(r[1]) - float(r[0]) for r in eq.param_ranges.values()]
                y = _ops.eq_dynamics_norm(x, eq_params, _functional._PEQ_TYPES, elo, espan, float(self.sample_rate), ctl, 0, 1e-8,
                                          None if flags is None else flags[0:1])
            else:
                # fused de-normalise + design; no gradient for x: the no-gx kernel
                y = eq.process_normalized(x, eq_params) if flags is None else eq.process_normalized(x, eq_params, _range_flag=flags[0:1])
                y = _ops.dynamics_ctl(y, 0, float(self.sample_rate), 1e-8, 0, ctl)
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


class ChainModule(torch.nn.Module):
    """The same chain as a torch.nn.Module built from torch.ops.dasp.* only (csrc/torch_ext): four op calls, no host read-back, no Python
    branching on tensors - `torch.compile(module, fullgraph=True)` captures it without a graph break, and in eager mode the backward pass
    runs entirely in C++. The reverb's noise is generated on the device from `noise_seed` plus the `seed_offset` buffer (bump it once per
    step: `module.seed_offset.add_(1)`). The [0, 1] contract of process_normalized is the caller's here (sigmoid heads satisfy it by
    construction; `StyleTransferChain` is the variant that checks), so no decay bound is vouched for and the filter bank decides its route
    per item. float32 tensors on one ROCm device."""

    def __init__(self, sample_rate, num_samples=65536, num_bandpass_taps=1023, noise_seed=0, device="cuda"):
        super().__init__()
        from . import _torch_ops, ops as _ops
        if not _torch_ops.load():
            raise _ops._lib.DaspHipError("ChainModule needs csrc/libdasp_torch.so (python -m dasp_pytorch_amd.csrc.build)")
        ref = StyleTransferChain(sample_rate)
        self.sample_rate, self.num_samples, self.taps, self.noise_seed = float(sample_rate), int(num_samples), int(num_bandpass_taps), int(noise_seed)
        self.types = list(_functional._PEQ_TYPES)
        self.eq_lo = [float(r[0]) for r in ref.equalizer.param_ranges.values()]
        self.eq_span = [float(r[1]) - float(r[0]) for r in ref.equalizer.param_ranges.values()]
        lo, span = ref._tables()
        self.lo, self.span = [float(v) for v in lo], [float(v) for v in span]
        dev = torch.device(device)
        filters = _functional._device_filterbank(self.taps, self.sample_rate, dev)
        import ctypes
        sizes = (ctypes.c_long * 14)()
        _ops.check(_ops._lib.lib().dasp_reverb_sizes(1, 4096, self.num_samples, self.taps, filters.shape[0], sizes), "dasp_reverb_sizes")
        with torch.cuda.device(dev):
            self.register_buffer("fspec", _ops._filter_spectrum(filters, filters.shape[0], self.taps, sizes[4], dev).clone(), persistent=False)
        self.bands = int(filters.shape[0])
        self.register_buffer("seed_offset", torch.zeros(1, dtype=torch.int64, device=dev), persistent=False)

    def forward(self, x, eq_params, comp_params, reverb_params, gain_params):
        d = torch.ops.dasp
        ctl, gains, decays, mix = d.chain_controls(comp_params, reverb_params, gain_params, self.lo, self.span)
        y = d.parametric_eq_norm(x, eq_params, self.sample_rate, self.types, self.eq_lo, self.eq_span)
        y = d.dynamics_ctl(y, ctl, 0, self.sample_rate, 1e-8, 0)
        return d.reverb(y, None, self.fspec, gains, decays, mix, self.num_samples, self.taps, self.bands, self.noise_seed, self.seed_offset, 0.0)
