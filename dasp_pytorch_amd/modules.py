"""Processor wrappers: the normalised-parameter API of dasp_pytorch.modules on top of the MI355X
functional layer (reference: dasp_pytorch/modules.py:21-231).

Same class names, constructor arguments, `param_ranges` (name -> (min, max), in the order of the
columns of the parameter tensor), `num_params`, `process`, `process_normalized`,
`extract_param_dict` and `denormalize_param_dict`. Differences, all deliberate:

* the range check of `process_normalized` is one min/max reduction over the whole (bs, P) tensor and
  one host sync *before* the processor's kernels are queued (so the host never waits for them), instead of the
  reference's two syncs per parameter (modules.py:83: 50 syncs for an EQ -> compressor -> reverb -> gain chain,
  each a GPU pipeline drain); the ValueError still names the offending parameter;
* de-normalisation is one affine op on the (bs, P) tensor, then column views;
* `Distortion` works: the reference's has no `sample_rate` attribute and names its parameter
  `gain_db` although `functional.distortion` takes `drive_db`, so `process_normalized` raises
  there (modules.py:110-121). Here the parameter is `drive_db`; the constructor keeps the reference's
  positional order (min_gain_db, max_gain_db) with `sample_rate` as an optional third argument;
* `Expander` exists (the reference's functional.expander is a stub).
"""
from typing import Dict

import atexit
import contextlib
import functools
import sys
import threading
import weakref

import torch

from . import functional as F


def denormalize(norm_val, max_val, min_val):
    return (norm_val * (max_val - min_val)) + min_val


def normalize(val, min_val, max_val):
    return (val - min_val) / (max_val - min_val)


_validated = threading.local()


@contextlib.contextmanager
def already_validated(enforced=True):
    """Inside this block `Processor._check_range` is a no-op: for callers that checked the parameter tensors of several processors with one
    reduction and one sync (chain.StyleTransferChain) before calling them. enforced=False: the caller only SUBMITTED a deferred check -
    nothing has vouched for this call's values yet, so promises derived from the range (the reverb's decay bound) stay withdrawn."""
    prev = (getattr(_validated, "on", False), getattr(_validated, "enforced", False))
    _validated.on, _validated.enforced = True, bool(enforced)
    try:
        yield
    finally:
        _validated.on, _validated.enforced = prev


def check_unit_range(param_tensor: torch.Tensor, names):
    """ValueError naming the first column of the (bs, P) tensor with an entry outside [0, 1] (reference: modules.py:83-84, same message).
    One min/max reduction, one two-element read-back; columns are only looked at when the check fails."""
    p = param_tensor.detach()
    if p.numel() == 0:
        return
    # NaNs pass, as they do the reference's (p < 0).any() / (p > 1).any() - but they must not hide an out-of-range value beside them
    lo, hi = torch.stack(torch.aminmax(torch.nan_to_num(p, nan=0.5))).tolist()
    if lo < 0 or hi > 1:
        bad = ((p < 0) | (p > 1)).any(dim=0)
        raise ValueError(f"Parameter {list(names)[int(torch.nonzero(bad)[0])]} of is out of range.")


# Deferred checks look at a call's numbers one call late; the LAST call of a run has no next call. Every checker with something pending is
# looked at once more when the interpreter exits (an offence found there is written to stderr - an exception cannot unwind into user code
# any more), and a call that starts under a stream capture - where nothing may wait - looks at a pending result only if it has already
# arrived (round 5, advisor).
_PENDING_CHECKERS = weakref.WeakSet()


def _flush_pending_at_exit():
    for chk in list(_PENDING_CHECKERS):
        try:
            chk.flush()
        except ValueError as e:
            sys.stderr.write(f"dasp_pytorch_amd: {e} [found at interpreter exit: the offending call was the last one of the run]\n")
        except Exception:
            pass                                   # (the device may be gone already)


atexit.register(_flush_pending_at_exit)


def _may_wait():
    return not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())


class _DeferredRangeCheck:
    """validate_range = "deferred": the [0, 1] check of process_normalized without the host waiting for the device. Each call queues ONE
    per-column min / max reduction and an asynchronous copy of its 2 P numbers into pinned host memory, and looks at the numbers of the
    PREVIOUS call - whose reduction was queued in front of that call's kernels and is long done - so a value outside [0, 1] still raises the
    reference's ValueError (modules.py:83-84, the parameter named from the host copy), one call late, and the host keeps running ahead of the
    GPU. Nothing of the parameter tensor is retained (round 4, advisor: the re-scan of a tensor that an optimizer had updated in place in
    the meantime could find no offender). `flush()` (or the next call) collects the last one."""

    def __init__(self):
        self.pending = None
        self.host = None

    def submit(self, param_tensor, names):
        self.flush()
        p = param_tensor.detach()
        if p.numel() == 0:
            return
        if not p.is_cuda:
            check_unit_range(p, names)
            return
        # (2, P) min / max per column with NaNs set aside: a NaN passes the reference's (p < 0).any() / (p > 1).any(), but a column that holds
        # a NaN AND an out-of-range value must still raise - aminmax alone would return NaN for it (round 5, advisor)
        mm = torch.stack((torch.nan_to_num(p, nan=0.5).amin(dim=0), torch.nan_to_num(p, nan=0.5).amax(dim=0))).to(torch.float32)
        if self.host is None or self.host.shape != mm.shape:
            self.host = torch.empty(mm.shape, dtype=torch.float32, pin_memory=True)
        self.host.copy_(mm, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending = (ev, list(names))
        _PENDING_CHECKERS.add(self)

    def flush(self):
        if self.pending is None:
            return
        ev, names = self.pending
        if not _may_wait() and not ev.query():
            return                                 # under a capture: looked at when it has arrived
        self.pending = None
        ev.synchronize()
        lo, hi = self.host[0].tolist(), self.host[1].tolist()
        for name, a, b in zip(names, lo, hi):
            if a < 0 or b > 1:
                raise ValueError(f"Parameter {name} of is out of range. (found by the deferred check: in the previous call)")


class _FlagRangeCheck:
    """validate_range = "deferred" on the fused GPU paths (round 5): the kernels that read the normalised parameters anyway - the EQ's design
    kernel, the chain's control kernel - OR a bit per offending column into persistent device words (C ABI: the `flag` argument of
    dasp_peq_forward_norm / dasp_chain_controls; torch.ops.dasp.*: `range_flag`), so the check costs the step no launch at all. Per call the
    host does one asynchronous copy of the words into pinned memory and one event record, and looks at the PREVIOUS call's copy: a value
    outside [0, 1] raises the reference's ValueError (modules.py:83-84, the first offending parameter named), one call late, exactly as
    _DeferredRangeCheck does with its five torch ops per call. The words are sticky until an error has been reported."""

    def __init__(self, nwords):
        self.nwords = nwords
        self.dev = None
        self.host = None
        self.pending = None

    def words(self, device):
        """The device words (int32, zero when created); word i as a 1-element view: words(dev)[i:i + 1]."""
        if self.dev is None or self.dev.device != device:
            self.flush()
            self.dev = torch.zeros(self.nwords, dtype=torch.int32, device=device)
            self.host = torch.zeros(self.nwords, dtype=torch.int32, pin_memory=True)
        return self.dev

    def begin(self):
        """Before a call queues anything: raise what the previous call found."""
        self.flush()

    def end(self, names_per_word):
        """After the call's kernels are queued: the words on their way to the host, for the next call to look at."""
        self.host.copy_(self.dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending = (ev, names_per_word)
        _PENDING_CHECKERS.add(self)

    def flush(self):
        if self.pending is None:
            return
        ev, names_per_word = self.pending
        if not _may_wait() and not ev.query():
            return                                 # under a capture: looked at when it has arrived
        self.pending = None
        ev.synchronize()
        for word, names in zip(self.host.tolist(), names_per_word):
            word &= 0xFFFFFFFF
            if word:
                self.dev.zero_()                      # reported: the words start afresh
                bit = (word & -word).bit_length() - 1
                raise ValueError(f"Parameter {names[bit]} of is out of range. (found by the deferred check: in the previous call)")


class Processor:
    """Base class with the reference's contract (modules.py:21-91): a subclass sets `sample_rate`, `process_fn` and `param_ranges`
    (name -> (min, max), in the order of the columns of the parameter tensor) - nothing else is required, so processors written against
    the reference work unchanged. The affine de-normalisation tables are derived from the *current* `param_ranges` on every call
    (cached per (ranges, device, dtype)), so editing the ranges after construction takes effect, as it does in the reference."""
    sample_rate = None
    process_fn = None
    param_ranges: Dict[str, tuple] = {}
    # The reference validates every normalised parameter on every call (modules.py:83-84: two host syncs per parameter). Here it is one
    # min/max read back per call, before the call's kernels are queued - still a sync with whatever was queued earlier. A training loop
    # whose controls come out of a sigmoid can switch it off per processor (`proc.validate_range = False`) or for the class; inside a
    # HIP-graph capture it is skipped in any case. `validate_range = "deferred"` keeps the check without the wait: every call reads the
    # PREVIOUS call's min / max (asynchronous copy to pinned memory), so the error is raised one call late (_DeferredRangeCheck).
    validate_range = True

    def __init__(self):
        pass

    @property
    def num_params(self):
        return len(self.param_ranges)

    @num_params.setter
    def num_params(self, value):      # the reference's subclasses assign it in __init__ (modules.py:107); accepted and ignored
        pass

    def _affine(self, ref: torch.Tensor):
        """(min, max - min) of every parameter as two (P,) tensors on ref's device / dtype."""
        cache = self.__dict__.setdefault("_affine_cache", {})
        key = (tuple(self.param_ranges.items()), ref.device, ref.dtype)
        hit = cache.get(key)
        if hit is None:
            capturing = ref.is_cuda and torch.cuda.is_current_stream_capturing()
            lo = torch.tensor([float(r[0]) for r in self.param_ranges.values()], dtype=torch.float64)
            span = torch.tensor([float(r[1]) - float(r[0]) for r in self.param_ranges.values()], dtype=torch.float64)
            hit = (lo.to(device=ref.device, dtype=ref.dtype), span.to(device=ref.device, dtype=ref.dtype))
            if not capturing:                  # memory allocated while a HIP graph is being captured belongs to that graph
                if len(cache) >= 8:
                    cache.clear()
                cache[key] = hit
        return hit

    def process_normalized(self, x: torch.Tensor, param_tensor: torch.Tensor):
        """Run the processor with parameters normalised to [0, 1], one row per batch item, columns in the
        order of `param_ranges` (reference: modules.py:25-51)."""
        if param_tensor.shape[1] != len(self.param_ranges):
            raise ValueError(
                f"Parameter tensor has {param_tensor.shape[1]} parameters, but processor has {len(self.param_ranges)} parameters.")
        self._check_range(param_tensor)
        lo, span = self._affine(param_tensor)
        denorm = param_tensor * span + lo                      # one op for all parameters
        kwargs = {name: denorm[:, i] for i, name in enumerate(self.param_ranges)}
        return self.process_fn(x, self.sample_rate, **kwargs)

    def process(self, x: torch.Tensor, *args):
        return self.process_fn(x, *args)

    def _check_range(self, param_tensor: torch.Tensor):
        # a HIP-graph capture cannot read a result back on the host: inside one the [0, 1] check is skipped (validate the
        # controls once in eager mode; a sigmoid head, as in the reference's models, satisfies it by construction)
        if not self.validate_range or getattr(_validated, "on", False) or (param_tensor.is_cuda and torch.cuda.is_current_stream_capturing()):
            return
        if self.validate_range == "deferred":
            self._deferred().submit(param_tensor, self.param_ranges)
            return
        check_unit_range(param_tensor, self.param_ranges)

    def _deferred(self):
        d = self.__dict__.get("_deferred_check")
        if d is None:
            d = self.__dict__["_deferred_check"] = _DeferredRangeCheck()
        return d

    def _flags(self, nwords=1):
        d = self.__dict__.get("_flag_check")
        if d is None:
            d = self.__dict__["_flag_check"] = _FlagRangeCheck(nwords)
        return d

    def flush_range_check(self):
        """validate_range = "deferred": raise now if the last call's parameters were outside [0, 1] (waits for that call's reduction)."""
        for key in ("_deferred_check", "_flag_check"):
            d = self.__dict__.get(key)
            if d is not None:
                d.flush()

    def _range_is_enforced(self):
        """True when the [0, 1] check of process_normalized actually RAN for this call: the caller already did it (chain.StyleTransferChain
        with the blocking check), or this processor's blocking check is on and the stream is not being captured into a HIP graph. A deferred
        check has only been submitted - this call's values are looked at one call later - and inside a capture `_check_range` is skipped:
        in both cases nothing vouches for the values the kernels will see, promises derived from the range (the reverb's decay bound) are
        withdrawn and the kernels decide per item."""
        if getattr(_validated, "on", False):
            return bool(getattr(_validated, "enforced", False))
        if self.validate_range == "deferred" or not self.validate_range:
            return False
        return not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())

    def extract_param_dict(self, param_tensor: torch.Tensor):
        if param_tensor.shape[1] != len(self.param_ranges):
            raise ValueError(
                f"Parameter tensor has {param_tensor.shape[1]} parameters, but processor has {len(self.param_ranges)} parameters.")
        return {name: param_tensor[:, i] for i, name in enumerate(self.param_ranges)}

    def denormalize_param_dict(self, param_dict: dict):
        """Parameters on [0, 1] -> the processor's physical ranges (reference: modules.py:70-91)."""
        out = {}
        for name, t in param_dict.items():
            if t.min() < 0 or t.max() > 1:
                raise ValueError(f"Parameter {name} of is out of range.")
            lo, hi = self.param_ranges[name]
            out[name] = denormalize(t, hi, lo)
        return out


class Gain(Processor):
    def __init__(self, sample_rate: int, min_gain_db: float = -24.0, max_gain_db: float = 24.0):
        super().__init__()
        self.sample_rate = sample_rate
        self.process_fn = F.gain
        self.param_ranges = {"gain_db": (min_gain_db, max_gain_db)}


class Distortion(Processor):
    """The reference's constructor order (min_gain_db, max_gain_db; modules.py:110-121). Its Distortion cannot run - no `sample_rate`
    attribute and a parameter named `gain_db` where functional.distortion takes `drive_db` - so here the
    parameter is `drive_db` and `sample_rate` is an optional trailing argument (functional.distortion ignores it)."""

    def __init__(self, min_gain_db: float = 0.0, max_gain_db: float = 24.0, sample_rate: int = None):
        super().__init__()
        self.sample_rate = sample_rate
        self.process_fn = F.distortion
        self.param_ranges = {"drive_db": (min_gain_db, max_gain_db)}


def _eq_ranges(sample_rate, g, q):
    top = (sample_rate // 2) - 1000
    cut = {"low_shelf": (20, 2000), "band0": (80, 2000), "band1": (2000, 8000), "band2": (8000, 12000), "band3": (12000, top),
           "high_shelf": (4000, top)}
    ranges = {}
    for band, c in cut.items():
        ranges[f"{band}_gain_db"] = g
        ranges[f"{band}_cutoff_freq"] = c
        ranges[f"{band}_q_factor"] = q
    return ranges


_EQ_NAMES = list(_eq_ranges(44100, (0, 1), (0, 1)))


class ParametricEQ(Processor):
    def __init__(self, sample_rate: int, min_gain_db: float = -20.0, max_gain_db: float = 20.0, min_q_factor: float = 0.1,
                 max_q_factor: float = 6.0):
        super().__init__()
        self.sample_rate = sample_rate
        self.process_fn = F.parametric_eq
        self.param_ranges = _eq_ranges(sample_rate, (min_gain_db, max_gain_db), (min_q_factor, max_q_factor))

    def _fused_ok(self, x, param_tensor):
        return (self.process_fn is F.parametric_eq and x.is_cuda and x.dtype is torch.float32 and x.dim() == 3 and param_tensor.dim() == 2
                and param_tensor.shape[1] == 18 and param_tensor.shape[0] in (1, x.shape[0]) and list(self.param_ranges) == _EQ_NAMES
                and param_tensor.is_floating_point() and param_tensor.device == x.device)

    def process_normalized(self, x: torch.Tensor, param_tensor: torch.Tensor, _range_flag: torch.Tensor = None):
        """As Processor.process_normalized; float32 audio on the GPU takes the fused op (ops.ParametricEQNormFunction): de-normalisation,
        filter design inside the design kernel, gradients returned w.r.t. `param_tensor` itself. Anything else (float64,
        a replaced process_fn, renamed ranges) goes through the generic path. `_range_flag` (chain.StyleTransferChain's deferred check):
        the device word the design kernel reports out-of-range columns into."""
        if not self._fused_ok(x, param_tensor):
            if _range_flag is not None:               # the caller counted on the kernel's check: do the blocking one instead
                check_unit_range(param_tensor, self.param_ranges)
            return super().process_normalized(x, param_tensor)
        from .ops import parametric_eq_norm
        lo = [float(r[0]) for r in self.param_ranges.values()]
        span = [float(r[1]) - float(r[0]) for r in self.param_ranges.values()]
        deferred = (_range_flag is None and self.validate_range == "deferred" and not getattr(_validated, "on", False)
                    and not torch.cuda.is_current_stream_capturing())
        if deferred:
            # the design kernel itself flags the columns outside [0, 1]; the word is read back one call late (_FlagRangeCheck)
            fc = self._flags()
            fc.begin()
            with torch.cuda.device(x.device):
                y = parametric_eq_norm(x, param_tensor, float(self.sample_rate), F._PEQ_TYPES, lo, span, fc.words(x.device))
                fc.end([list(self.param_ranges)])
            return y
        # the blocking check runs before the kernels are queued (one small reduction + read-back)
        self._check_range(param_tensor)
        return parametric_eq_norm(x, param_tensor, float(self.sample_rate), F._PEQ_TYPES, lo, span, _range_flag)


class _Dynamics(Processor):
    def __init__(self, fn, sample_rate: int, min_threshold_db: float = -60.0, max_threshold_db: float = 0.0, min_ratio: float = 1.0,
                 max_ratio: float = 20.0, min_attack_ms: float = 5.0, max_attack_ms: float = 100.0, min_release_ms: float = 5.0,
                 max_release_ms: float = 100.0, min_knee_db: float = 0.0, max_knee_db: float = 12.0, min_makeup_gain_db: float = 0.0,
                 max_makeup_gain_db: float = 12.0):
        Processor.__init__(self)
        self.sample_rate = sample_rate
        self.process_fn = fn
        self.param_ranges = {
            "threshold_db": (min_threshold_db, max_threshold_db), "ratio": (min_ratio, max_ratio),
            "attack_ms": (min_attack_ms, max_attack_ms), "release_ms": (min_release_ms, max_release_ms),
            "knee_db": (min_knee_db, max_knee_db), "makeup_gain_db": (min_makeup_gain_db, max_makeup_gain_db)}


_DYN_NAMES = ["threshold_db", "ratio", "attack_ms", "release_ms", "knee_db", "makeup_gain_db"]


def _dynamics_process_normalized(self, mode, x, param_tensor):
    """Compressor / Expander.process_normalized: float32 audio on the GPU hands the de-normalised (bs, 6) matrix to the kernels as one
    tensor (ops.DynamicsMatrixFunction) instead of six column views that are stacked again one call further down."""
    fused = (x.is_cuda and x.dtype is torch.float32 and x.dim() == 3 and param_tensor.dim() == 2 and param_tensor.shape == (x.shape[0], 6)
             and list(self.param_ranges) == _DYN_NAMES and param_tensor.is_floating_point())
    if not fused:
        return Processor.process_normalized(self, x, param_tensor)
    self._check_range(param_tensor)
    lo, span = self._affine(param_tensor)
    return F._dynamics_from_matrix(mode, x, self.sample_rate, param_tensor * span + lo)


class Compressor(_Dynamics):
    """Positional order of the reference's constructor (modules.py:159-187)."""

    def __init__(self, sample_rate: int, min_threshold_db: float = -60.0, max_threshold_db: float = 0.0, min_ratio: float = 1.0,
                 max_ratio: float = 20.0, min_attack_ms: float = 5.0, max_attack_ms: float = 100.0, min_release_ms: float = 5.0,
                 max_release_ms: float = 100.0, min_knee_db: float = 0.0, max_knee_db: float = 12.0, min_makeup_gain_db: float = 0.0,
                 max_makeup_gain_db: float = 12.0):
        super().__init__(F.compressor, sample_rate, min_threshold_db, max_threshold_db, min_ratio, max_ratio, min_attack_ms, max_attack_ms,
                         min_release_ms, max_release_ms, min_knee_db, max_knee_db, min_makeup_gain_db, max_makeup_gain_db)

    def process_normalized(self, x: torch.Tensor, param_tensor: torch.Tensor):
        if self.process_fn is not F.compressor:
            return super().process_normalized(x, param_tensor)
        return _dynamics_process_normalized(self, 0, x, param_tensor)


class Expander(_Dynamics):
    def __init__(self, sample_rate: int, min_threshold_db: float = -60.0, max_threshold_db: float = 0.0, min_ratio: float = 1.0,
                 max_ratio: float = 20.0, min_attack_ms: float = 5.0, max_attack_ms: float = 100.0, min_release_ms: float = 5.0,
                 max_release_ms: float = 100.0, min_knee_db: float = 0.0, max_knee_db: float = 12.0, min_makeup_gain_db: float = 0.0,
                 max_makeup_gain_db: float = 12.0):
        super().__init__(F.expander, sample_rate, min_threshold_db, max_threshold_db, min_ratio, max_ratio, min_attack_ms, max_attack_ms,
                         min_release_ms, max_release_ms, min_knee_db, max_knee_db, min_makeup_gain_db, max_makeup_gain_db)

    def process_normalized(self, x: torch.Tensor, param_tensor: torch.Tensor):
        if self.process_fn is not F.expander:
            return super().process_normalized(x, param_tensor)
        return _dynamics_process_normalized(self, 1, x, param_tensor)


_REV_NAMES = [f"band{i}_gain" for i in range(12)] + [f"band{i}_decay" for i in range(12)] + ["mix"]


class NoiseShapedReverb(Processor):
    def __init__(self, sample_rate, min_band_gain: float = 0.0, max_band_gain: float = 1.0, min_band_decay: float = 0.0,
                 max_band_decay: float = 1.0, min_mix: float = 0.0, max_mix: float = 1.0, num_samples: int = 65536,
                 num_bandpass_taps: int = 1023, device_noise: bool = False, noise_seed_offset: torch.Tensor = None, noise_seed: int = None):
        """The last four arguments are additions to the reference's constructor (defaults = the reference's behaviour):
        impulse-response length, filter-bank taps, generating the noise on the GPU instead of drawing it from the global CPU generator, and
        (with device_noise) a 1-element int64 device tensor added to the noise seed when the kernels run - for HIP-graph replays - and a fixed
        base seed instead of one draw per call from torch's CPU generator; see functional.noise_shaped_reverberation."""
        super().__init__()
        self.sample_rate = sample_rate
        self.process_fn = functools.partial(F.noise_shaped_reverberation, num_samples=num_samples, num_bandpass_taps=num_bandpass_taps,
                                            device_noise=device_noise, noise_seed_offset=noise_seed_offset, noise_seed=noise_seed)
        self.param_ranges = {f"band{i}_gain": (min_band_gain, max_band_gain) for i in range(12)}
        self.param_ranges.update({f"band{i}_decay": (min_band_decay, max_band_decay) for i in range(12)})
        self.param_ranges["mix"] = (min_mix, max_mix)
        self._rev_kwargs = dict(num_samples=num_samples, num_bandpass_taps=num_bandpass_taps, device_noise=device_noise,
                                noise_seed_offset=noise_seed_offset, noise_seed=noise_seed)
        self._rev_fn = self.process_fn

    def process_normalized(self, x: torch.Tensor, param_tensor: torch.Tensor):
        """As Processor.process_normalized; float32 audio on the GPU passes the de-normalised (bs, 25) tensor on as three slices (band
        gains, band decays, mix) instead of 25 column views that functional.noise_shaped_reverberation would stack again."""
        fused = (self.process_fn is self._rev_fn and x.is_cuda and x.dtype is torch.float32 and x.dim() == 3 and x.shape[1] <= 2
                 and param_tensor.dim() == 2 and param_tensor.shape == (x.shape[0], 25) and param_tensor.is_floating_point()
                 and list(self.param_ranges) == _REV_NAMES)
        if not fused:
            return super().process_normalized(x, param_tensor)
        self._check_range(param_tensor)
        lo, span = self._affine(param_tensor)
        d = param_tensor * span + lo
        # the validated parameter range bounds the decays: the filter bank then knows which of its two routes every item takes
        return F._reverb_from_matrices(x, self.sample_rate, d[:, :12], d[:, 12:24], d[:, 24], decay_bound=self._decay_bound(), **self._rev_kwargs)

    def _decay_bound(self):
        """Largest band decay the validated parameter range allows (0 = no promise: the range check is switched off)."""
        if not self._range_is_enforced() or list(self.param_ranges) != _REV_NAMES:
            return 0.0
        return max(abs(float(v)) for i in range(12) for v in self.param_ranges[f"band{i}_decay"])
