"""torch.autograd.Function wrappers over the C ABI (include/dasp_hip.h) through ctypes.

The product's autograd binding is the PyTorch extension (csrc/torch_ext -> torch.ops.dasp.*; ops.py routes to it). This module is
 - the binding of what the extension does not register: biquad, the per-sample distortion, the stereo utilities, dynamics on a matrix
   of controls, and every float64 route (ops64.py);
 - the SECOND binding of what it does register (sosfilt, parametric_eq, gain, distortion, dynamics, the chain's normalised ops, the
   reverb): the GPU tests compare the two, bench.py's per-call HIP events go through it, config.plan.torch_ops = False selects it,
   and it owns the dtype / device / shape error messages of the calls the extension does not take.
PyTorch is used for device memory, streams and autograd plumbing only; every number is produced by the HIP kernels in csrc/. Nothing here
falls back to torch math on a missing library or a CPU tensor.
"""
import ctypes

import torch
from torch.autograd.function import once_differentiable

from . import _lib, config
from ._lib import call, check, ptr, stream

FILTER_TYPES = {"peaking": 0, "low_shelf": 1, "high_shelf": 2, "low_pass": 3, "high_pass": 4}


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def _pad_sections(S):
    """Smallest compiled section count >= S (kernels exist for 2/4/6/8 sections)."""
    for s in (2, 4, 6, 8):
        if S <= s:
            return s
    raise ValueError("more than 8 sections per call: chain calls (see signal.sosfilt_via_fsm)")


def _segment_tiles(rows, N, generic=False):
    """Tiles per segment for the segmented-row kernels, 0 = one workgroup per row (dasp_hip.h, "Few rows").
    A row is one workgroup, so few rows (the reference's training batches are 8-32 items: examples/style_transfer.py:403,
    auto_eq.py:231) leave most of the 256 CUs idle; the segmented path takes 2-4x less GPU time there (16 x 2 x 131072: forward
    0.078 -> 0.032 ms, backward 0.177 -> 0.047 ms) for four more kernel launches per call, all issued by the same C call. It is taken
    whenever the library's planner proposes a cut (at most 128 rows and at least 16 tiles per row; above that one workgroup per row runs at
    twice the waves per row up to 256 rows and is faster), eager or captured.
    config.plan.sos_segment = False: never; config.plan.sos_segment_tiles = <power of two> fixes the segment length."""
    if not config.plan.sos_segment:
        return 0
    if config.plan.sos_segment_tiles:
        return int(config.plan.sos_segment_tiles)
    # generic: a cascade given by its coefficients (no design launch per call: its segmented rows keep the pre-pass launches, five launches
    # per step) - there segments stop paying above 64 rows (profiles/r04/seg_crossover.log); the designed paths go up to the planner's 128
    return 0 if generic and rows > 64 else int(_lib.lib().dasp_sos_segment_tiles(rows, N))


def _round64(n):
    return (int(n) + 63) & ~63


class _SosWork:
    """Device work buffers of one filter application: one fp32 block (tables, saved chunk states, partial sums, segment scratch) and
    one fp64 block (design side table, segment transition matrices) - two allocations per call instead of one per buffer."""

    def __init__(self, Bs, S, x, need_grad, generic=False):
        L = _lib.lib()
        B, C, N = x.shape
        self.Bs, self.S = Bs, S
        self.tseg = _segment_tiles(B * C, N, generic)
        self.G = int(L.dasp_sos_segments(N, self.tseg))
        n_tab = _round64(Bs * L.dasp_sos_table_floats(S))
        n_car = _round64(L.dasp_sos_carry_floats(B * C, N, S)) if need_grad else 0
        n_par = _round64(L.dasp_sos_partial_floats(B * C * self.G, S)) if need_grad else 0
        n_seg = _round64(L.dasp_sos_seg_floats(B * C, N, S, self.tseg)) if self.tseg else 0
        f32 = torch.empty(n_tab + n_car + n_par + n_seg, dtype=torch.float32, device=x.device)
        self.tab, self.carries = f32[:n_tab], (f32[n_tab:n_tab + n_car] if need_grad else None)
        self.partials = f32[n_tab + n_car:n_tab + n_car + n_par] if need_grad else None
        self.segbuf = f32[n_tab + n_car + n_par:] if self.tseg else None
        n_dt = Bs * L.dasp_sos_dtab_doubles(S)
        f64 = torch.empty(n_dt + (Bs * L.dasp_sos_segtab_doubles(S) if self.tseg else 0), dtype=torch.float64, device=x.device)
        self.dtab, self.segtab = f64[:n_dt], (f64[n_dt:] if self.tseg else None)

    def forward(self, x):
        """Cascade from tables that are already filled (dasp_sos_prepare)."""
        B, C, N = x.shape
        y = torch.empty_like(x)
        if self.tseg:
            call("dasp_sos_segment_prepare", ptr(self.dtab), self.Bs, self.S, self.tseg, ptr(self.segtab), stream())
            call("dasp_sosfilt_forward_seg", ptr(self.tab), ptr(self.segtab), self.Bs, ptr(x), ptr(y), ptr(self.carries), ptr(self.segbuf),
                 B, C, N, self.S, self.tseg, stream())
        else:
            call("dasp_sosfilt_forward", ptr(self.tab), self.Bs, ptr(x), ptr(y), ptr(self.carries), B, C, N, self.S, stream())
        return y

    def backward(self, x, gy, mode, designed, need_gx, need_gc):
        """Adjoint cascade and coefficient / control gradients; (gx or None, gout or None)."""
        B, C, N = x.shape
        gx = torch.empty_like(x) if need_gx else None
        gout = None
        if need_gc:
            shape = (B, self.S, 6) if mode == 0 else (B, self.S, 3) if mode == 1 else (3 * self.S, B)
            gout = torch.empty(shape, dtype=torch.float32, device=x.device)
        part = self.partials if need_gc else None
        if _lib.timers.enabled and not self.tseg:      # bench.py's per-kernel HIP events: the two launches as separate entry points
            call("dasp_sosfilt_backward_ex", ptr(self.tab), self.Bs, ptr(x), ptr(gy), ptr(self.carries), ptr(gx), ptr(part), B, C, N, self.S, stream())
            if need_gc:
                call("dasp_sos_grad_finalize_ex", ptr(self.dtab), self.Bs, ptr(part), B, C, self.S, 1, mode, ptr(gout), stream())
        elif self.tseg and designed:
            # segmented rows of a designed cascade: pre-pass (+ chain), adjoint pass (+ finalize in its last workgroup per item) - two launches
            call("dasp_peq_backward", ptr(self.tab), ptr(self.dtab), self.Bs, ptr(x), ptr(gy), ptr(self.carries), ptr(gx), ptr(part), mode,
                 ptr(gout), B, C, N, self.S, self.tseg, ptr(self.segtab), ptr(self.segbuf), stream())
        elif self.tseg:
            call("dasp_sosfilt_backward_seg_ex", ptr(self.tab), ptr(self.segtab), self.Bs, ptr(x), ptr(gy), ptr(self.carries), ptr(gx),
                 ptr(part), ptr(self.segbuf), B, C, N, self.S, self.tseg, stream())
            if need_gc:
                call("dasp_sos_grad_finalize_ex", ptr(self.dtab), self.Bs, ptr(part), B, C, self.S, self.G, mode, ptr(gout), stream())
        else:
            call("dasp_sosfilt_backward_grads_ex", ptr(self.tab), ptr(self.dtab), self.Bs, ptr(x), ptr(gy), ptr(self.carries), ptr(gx),
                 ptr(part), mode, ptr(gout), B, C, N, self.S, stream())
        if need_gc and self.Bs == 1 and B != 1:
            gout = gout.sum(1 if mode == 2 else 0, keepdim=True)
        return gx, gout


class SosFiltFunction(torch.autograd.Function):
    """y = cascade of S biquads `sos` (Bs,S,6) applied to x (B,C,N); grads for x and sos."""

    @staticmethod
    def forward(ctx, sos, x):
        _lib.require_device(x, "x")
        _lib.require_device(sos, "sos")
        _lib.require_same_device(x, sos=sos)
        Bs, S, _ = sos.shape
        Sp = _pad_sections(S)
        ctx.dtypes = (sos.dtype, x.dtype)
        ctx.empty = x.numel() == 0
        if ctx.empty:
            ctx.shapes = (sos.shape, x.shape)
            return torch.empty_like(x)
        with torch.cuda.device(x.device):
            sos32 = _f32c(sos)
            if Sp != S:  # identity sections [1 0 0 1 0 0]
                pad = torch.zeros(Bs, Sp - S, 6, dtype=torch.float32, device=sos.device)
                pad[..., 0] = 1.0
                pad[..., 3] = 1.0
                sos32 = torch.cat([sos32, pad], 1).contiguous()
            x32 = _f32c(x)
            need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
            w = _SosWork(Bs, Sp, x32, need, generic=True)
            call("dasp_sos_prepare", ptr(sos32), Bs, Sp, ptr(w.tab), ptr(w.dtab), stream())
            y = w.forward(x32)
            if need:
                ctx.work, ctx.S = w, S
                ctx.save_for_backward(x32)
        return y.to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        if ctx.empty:
            return torch.zeros(ctx.shapes[0], dtype=ctx.dtypes[0], device=gy.device), torch.empty(ctx.shapes[1], dtype=ctx.dtypes[1], device=gy.device)
        (x32,) = ctx.saved_tensors
        with torch.cuda.device(x32.device):
            gx, gsos = ctx.work.backward(x32, _f32c(gy), 0, 0, ctx.needs_input_grad[1], ctx.needs_input_grad[0])
        return (gsos[:, :ctx.S].to(ctx.dtypes[0]) if gsos is not None else None, gx.to(ctx.dtypes[1]) if gx is not None else None)


class BiquadFunction(torch.autograd.Function):
    """signal.biquad: (gain_db, cutoff_freq, q_factor) with n values each -> (n, 6) fp64 rows [b0 b1 b2 1 a1 a2] (dasp_biquad_design);
    the backward contracts the in-kernel Jacobian with the incoming gradient (dasp_biquad_backward)."""

    @staticmethod
    def forward(ctx, gain_db, cutoff_freq, q_factor, sample_rate, ftype):
        _lib.require_device(gain_db, "gain_db")
        _lib.require_same_device(gain_db, cutoff_freq=cutoff_freq, q_factor=q_factor)
        n = gain_db.numel()
        ctx.meta = [(t.dtype, t.shape) for t in (gain_db, cutoff_freq, q_factor)]
        dev = gain_db.device
        ba = torch.empty(n, 6, dtype=torch.float64, device=dev)
        jac = torch.empty(n, 15, dtype=torch.float64, device=dev)
        if n:
            with torch.cuda.device(dev):
                g, f, q = (t.detach().reshape(-1).to(torch.float64).contiguous() for t in (gain_db, cutoff_freq, q_factor))
                call("dasp_biquad_design", ptr(g), ptr(f), ptr(q), n, int(ftype), float(sample_rate), ptr(ba), ptr(jac), stream())
        ctx.save_for_backward(jac)        # also for n == 0: the backward pass then returns empty gradients of the recorded shapes
        return ba

    @staticmethod
    @once_differentiable
    def backward(ctx, gba):
        (jac,) = ctx.saved_tensors
        n = jac.shape[0]
        gp = torch.zeros(n, 3, dtype=torch.float64, device=gba.device)
        if n:
            with torch.cuda.device(jac.device):
                call("dasp_biquad_backward", ptr(jac), ptr(gba.to(torch.float64).contiguous()), n, ptr(gp), stream())
        cols = gp.unbind(1)
        return tuple(c.reshape(shape).to(dt) for c, (dt, shape) in zip(cols, ctx.meta)) + (None, None)


class ParametricEQFunction(torch.autograd.Function):
    """Fused RBJ design (fp64, in-kernel) + cascade. `controls` are the 3*S per-item controls in the
    reference's argument order [gain_db, cutoff_freq, q_factor] per section, each with Bp elements.
    They enter as separate tensors and their gradients leave as contiguous rows of one (3S, Bp)
    buffer, so autograd neither builds a stack node nor launches 3S copy kernels. One C call per direction
    (dasp_peq_forward / dasp_peq_backward); the backward kernel is the variant torch.autograd's needs_input_grad asks for."""

    @staticmethod
    def forward(ctx, x, sample_rate, types, *controls):
        _lib.require_device(x, "x")
        L = _lib.lib()
        S = len(types)
        if not L.dasp_sos_supported_sections(S):
            raise ValueError(f"no kernel compiled for {S} sections")
        dev = x.device
        ctx.x_dtype = x.dtype
        ctx.ctl = [(c.dtype, c.shape) for c in controls]
        ctx.empty = x.numel() == 0
        if ctx.empty:
            return torch.empty_like(x)
        with torch.cuda.device(dev):
            # the usual case - a 1-D contiguous fp32 tensor on x's device - is used as it is (grad mode is off in here): 18 controls
            # through four no-op tensor calls each were a fifth of the host time of a step
            cols = [c if (c.dtype is torch.float32 and c.dim() == 1 and c.device == dev and c.is_contiguous())
                    else c.detach().reshape(-1).to(device=dev, dtype=torch.float32).contiguous() for c in controls]
            Bp = cols[0].numel()
            if any(c.numel() != Bp for c in cols):
                raise ValueError("parametric_eq controls must all have the same number of elements")
            x32 = _f32c(x)
            B, C, N = x32.shape
            need = any(ctx.needs_input_grad)
            w = _SosWork(Bp, S, x32, need)
            ctypes_types = (ctypes.c_int * S)(*types)
            rows = (ctypes.c_void_p * (3 * S))(*[c.data_ptr() for c in cols])        # read by the design kernel in place: no packing copy
            if _lib.timers.enabled and not w.tseg:   # bench.py's per-kernel HIP events: design and cascade as separate entry points
                call("dasp_peq_prepare_rows", rows, Bp, S, ctypes_types, float(sample_rate), ptr(w.tab), ptr(w.dtab), stream())
                y = w.forward(x32)
            else:
                y = torch.empty_like(x32)
                call("dasp_peq_forward", rows, Bp, S, ctypes_types, float(sample_rate), ptr(w.tab), ptr(w.dtab), ptr(x32), ptr(y),
                     ptr(w.carries), B, C, N, w.tseg, ptr(w.segtab), ptr(w.segbuf), stream())
            if need:
                ctx.work = w
                ctx.save_for_backward(x32)
        return y.to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        if ctx.empty:
            return (torch.empty_like(gy), None, None) + tuple(torch.zeros(shape, dtype=dt, device=gy.device) for dt, shape in ctx.ctl)
        (x32,) = ctx.saved_tensors
        need_gx, need_gc = ctx.needs_input_grad[0], any(ctx.needs_input_grad[3:])
        with torch.cuda.device(x32.device):
            gx, gpt = ctx.work.backward(x32, _f32c(gy), 2, 1, need_gx, need_gc)       # (3S, Bp): one contiguous gradient row per control tensor
        gcols = (None,) * len(ctx.ctl)
        if need_gc:
            rows = gpt.unbind(0)
            gcols = tuple((rows[i] if (dt is torch.float32 and shape == rows[i].shape) else rows[i].reshape(shape).to(dt)) if need else None
                          for i, ((dt, shape), need) in enumerate(zip(ctx.ctl, ctx.needs_input_grad[3:])))
        return (gx.to(ctx.x_dtype) if need_gx else None, None, None) + gcols


class ParametricEQNormFunction(torch.autograd.Function):
    """Processor.process_normalized for the EQ as one op (reference: dasp_pytorch/modules.py:25-91 + functional.py:118-272):
    the normalised (Bp, 3 S) tensor goes straight into the design kernel, which de-normalises it (lo + span * p in fp64), checks [0, 1] and
    builds the tables; the backward pass returns the gradient w.r.t. the normalised tensor. One tensor input instead of 3 S, no
    de-normalisation / slicing / stacking ops around the kernels. The [0, 1] check of the reference (modules.py:83) is the caller's
    (modules.check_unit_range, before anything is queued); the C entry point's own in-kernel flag word (dasp_hip.h) is not used from
    Python: reading it back would make the host wait for the forward kernel it has just queued."""

    @staticmethod
    def forward(ctx, x, pn, sample_rate, types, lo, span, range_flag=None):
        _lib.require_device(x, "x")
        _lib.require_same_device(x, param_tensor=pn)
        S = len(types)
        dev = x.device
        ctx.meta = (x.dtype, pn.dtype, pn.shape)
        ctx.empty = x.numel() == 0
        if ctx.empty:
            return torch.empty_like(x)
        with torch.cuda.device(dev):
            pn32 = _f32c(pn)
            Bp = pn32.shape[0]
            x32 = _f32c(x)
            B, C, N = x32.shape
            need = any(ctx.needs_input_grad)
            w = _SosWork(Bp, S, x32, need)
            y = torch.empty_like(x32)
            call("dasp_peq_forward_norm", ptr(pn32), Bp, S, (ctypes.c_int * S)(*types), float(sample_rate), (ctypes.c_double * (3 * S))(*lo),
                 (ctypes.c_double * (3 * S))(*span), ptr(range_flag), ptr(w.tab), ptr(w.dtab), ptr(x32), ptr(y), ptr(w.carries), B, C, N, w.tseg,
                 ptr(w.segtab), ptr(w.segbuf), stream())
            if need:
                ctx.work = w
                ctx.save_for_backward(x32)
        return y.to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xd, pd, pshape = ctx.meta
        if ctx.empty:
            return torch.empty_like(gy), torch.zeros(pshape, dtype=pd, device=gy.device), None, None, None, None, None
        (x32,) = ctx.saved_tensors
        need_gx, need_gp = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        with torch.cuda.device(x32.device):
            gx, gp = ctx.work.backward(x32, _f32c(gy), 1, 1, need_gx, need_gp)       # mode 1: (Bp, S, 3) = the layout of the (Bp, 3 S) tensor
        return (gx.to(xd) if need_gx else None, gp.reshape(pshape).to(pd) if need_gp else None, None, None, None, None, None)


class _ElementwiseFunction(torch.autograd.Function):
    """Shared plumbing of gain / distortion: y = f(x, ctl), ctl one dB value per batch item (gain)
    or per (b, c) row (distortion)."""
    FWD = BWD = None

    @classmethod
    def _run(cls, ctx, x, ctl):
        _lib.require_device(x, "x")
        B, C, N = x.shape
        ctx.meta = (x.dtype, ctl.dtype, ctl.shape)
        ctx.empty = x.numel() == 0
        if ctx.empty:
            return torch.empty_like(x)
        with torch.cuda.device(x.device):
            x32 = _f32c(x)
            c32 = ctl.detach().reshape(-1).to(device=x.device, dtype=torch.float32).contiguous()
            y = torch.empty_like(x32)
            call(cls.FWD, ptr(x32), ptr(c32), ptr(y), B, C, N, stream())
            ctx.save_for_backward(x32, c32)
        return y.to(x.dtype)

    @classmethod
    def _grad(cls, ctx, gy):
        xd, cd, cshape = ctx.meta
        if ctx.empty:
            return torch.empty_like(gy), torch.zeros(cshape, dtype=cd, device=gy.device)
        L = _lib.lib()
        x32, c32 = ctx.saved_tensors
        B, C, N = x32.shape
        with torch.cuda.device(x32.device):
            gx = torch.empty_like(x32)
            gctl = torch.empty_like(c32)
            partials = torch.empty(L.dasp_ew_partial_floats(B * C, N), dtype=torch.float32, device=x32.device)
            call(cls.BWD, ptr(x32), ptr(c32), ptr(_f32c(gy)), ptr(gx), ptr(gctl), ptr(partials), B, C, N, stream())
        return gx.to(xd), gctl.reshape(cshape).to(cd)


class GainFunction(_ElementwiseFunction):
    """y = x * 10^(gain_db/20); gain_db holds one value per batch item (functional.py:10-29)."""
    FWD, BWD = "dasp_gain_forward", "dasp_gain_backward"

    @staticmethod
    def forward(ctx, x, gain_db):
        return GainFunction._run(ctx, x, gain_db)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        return GainFunction._grad(ctx, gy)


class DistortionFunction(_ElementwiseFunction):
    """y = tanh(x * 10^(drive_db/20)); drive_db holds one value per (b, c) row (functional.py:65-78)."""
    FWD, BWD = "dasp_distortion_forward", "dasp_distortion_backward"

    @staticmethod
    def forward(ctx, x, drive_db):
        return DistortionFunction._run(ctx, x, drive_db)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        return DistortionFunction._grad(ctx, gy)


class DistortionSampleFunction(torch.autograd.Function):
    """y = tanh(x * 10^(drive_db/20)) with one drive value per sample: drive_db holds bs * chs * seq_len values (the other case the reference's
    drive_db.view(bs, chs, -1) accepts, functional.py:78)."""

    @staticmethod
    def forward(ctx, x, drive_db):
        _lib.require_device(x, "x")
        _lib.require_same_device(x, drive_db=drive_db)
        ctx.meta = (x.dtype, drive_db.dtype, drive_db.shape)
        ctx.empty = x.numel() == 0
        if ctx.empty:
            return torch.empty_like(x)
        with torch.cuda.device(x.device):
            x32 = _f32c(x)
            d32 = _f32c(drive_db).reshape(x.shape)
            y = torch.empty_like(x32)
            call("dasp_distortion_sample_forward", ptr(x32), ptr(d32), ptr(y), x32.numel(), stream())
            ctx.save_for_backward(x32, d32)
        return y.to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xd, dd, dshape = ctx.meta
        if ctx.empty:
            return torch.empty_like(gy), torch.zeros(dshape, dtype=dd, device=gy.device)
        x32, d32 = ctx.saved_tensors
        with torch.cuda.device(x32.device):
            gx = torch.empty_like(x32)
            gd = torch.empty_like(x32)
            call("dasp_distortion_sample_backward", ptr(x32), ptr(d32), ptr(_f32c(gy)), ptr(gx), ptr(gd), x32.numel(), stream())
        return gx.to(xd), gd.reshape(dshape).to(dd)


def _require_rows(x, t, ncols, name):
    """The dynamics kernels read `ncols` controls per batch item of x at t[b * ncols ...]: anything but a (bs, ncols) matrix would be read
    past its end. Same error as functional._dynamics (the reference's .view(-1, 1, 1) against a (bs, 1, seq_len) side chain does not
    broadcast a parameter batch of 1 either, functional.py:330-336)."""
    if t.dim() != 2 or t.shape[0] != x.shape[0] or t.shape[1] != ncols:
        raise RuntimeError(f"The size of tensor a ({t.shape[0] if t.dim() else 1}) must match the size of tensor b ({x.shape[0]}) at "
                           f"non-singleton dimension 0 ({name} must be ({x.shape[0]}, {ncols}), got {tuple(t.shape)})")


_DYN_COUNTERS = {}
_lib.on_failure.append(_DYN_COUNTERS.clear)          # a failed call may leave a count behind: the next call gets fresh zeros (advisor, r05)


def _dyn_counters(dev):
    """The completion counters of the segmented dynamics calls (4 ints per batch item, one buffer per (device, stream): 4 * 128 ints, and
    the segmented path is only taken up to 128 items - the size contract of dasp_hip.h: zero before the first use, every call returns
    the words it used to zero). Inside a HIP-graph capture a fresh zeroed buffer is used and not kept: it belongs to the graph's pool."""
    capturing = torch.cuda.is_current_stream_capturing()
    key = (dev.index, int(torch.cuda.current_stream(dev).cuda_stream))
    t = None if capturing else _DYN_COUNTERS.get(key)
    if t is None:
        t = torch.zeros(4 * 128, dtype=torch.int32, device=dev)          # (the segmented path is only taken below 128 items)
        if not capturing:
            if len(_DYN_COUNTERS) >= 16:
                _DYN_COUNTERS.clear()
            _DYN_COUNTERS[key] = t
    return t


def _dyn_forward(x, mode, sample_rate, eps, lookahead, ctl, need):
    """The compressor / expander kernels on ctl (B, 5) fp32 rows [threshold_db, ratio, attack_ms, knee_db, makeup_gain_db]; returns y (fp32)
    and what the backward pass needs."""
    L = _lib.lib()
    B, C, N = x.shape
    x32 = _f32c(x)
    y = torch.empty_like(x32)
    carries = torch.empty(L.dasp_dyn_carry_floats(B, N), dtype=torch.float32, device=x.device) if need else None
    lin = torch.empty(B, N, dtype=torch.float32, device=x.device) if lookahead > 0 else None
    # few items: every item is cut into segments that run as independent workgroups (dasp_hip.h, "Few batch items")
    tseg = 0 if not config.plan.dyn_segment else int(config.plan.dyn_segment_tiles or L.dasp_dyn_segment_tiles(B, N))
    segbuf = torch.empty(2 * B * L.dasp_dyn_segments(N, tseg), dtype=torch.float32, device=x.device) if tseg else None
    if tseg:
        call("dasp_dynamics_forward_seg", mode, ptr(x32), ptr(ctl), ptr(y), ptr(carries), ptr(lin), ptr(segbuf), B, C, N, float(sample_rate),
             float(eps), int(lookahead), tseg, ptr(_dyn_counters(x.device) if B <= 128 else None), stream())
    else:
        call("dasp_dynamics_forward", mode, ptr(x32), ptr(ctl), ptr(y), ptr(carries), ptr(lin), B, C, N, float(sample_rate),
             float(eps), int(lookahead), stream())
    saved = (x32, ctl, carries, lin if lin is not None else torch.empty(0, device=x.device))
    return y, saved, (mode, float(sample_rate), float(eps), int(lookahead), tseg)


def _dyn_backward(saved, cfg, gy):
    """gx (fp32) and gctl (B, 5) for the rows of ctl."""
    L = _lib.lib()
    x32, ctl, carries, lin = saved
    mode, sr, eps, look, tseg = cfg
    B, C, N = x32.shape
    gx = torch.empty_like(x32)
    gctl = torch.empty(B, 5, dtype=torch.float32, device=x32.device)
    G = int(L.dasp_dyn_segments(N, tseg))
    partials = torch.empty(L.dasp_dyn_partial_floats(B * G), dtype=torch.float32, device=x32.device)
    if tseg:
        segbuf = torch.empty(2 * B * G, dtype=torch.float32, device=x32.device)
        call("dasp_dynamics_backward_seg", mode, ptr(x32), ptr(ctl), ptr(_f32c(gy)), ptr(carries), ptr(lin if look > 0 else None),
             ptr(gx), ptr(gctl), ptr(partials), ptr(segbuf), B, C, N, sr, eps, look, tseg, ptr(_dyn_counters(x32.device) if B <= 128 else None),
             stream())
    else:
        call("dasp_dynamics_backward", mode, ptr(x32), ptr(ctl), ptr(_f32c(gy)), ptr(carries), ptr(lin if look > 0 else None),
             ptr(gx), ptr(gctl), ptr(partials), B, C, N, sr, eps, look, stream())
    return gx, gctl


class DynamicsFunction(torch.autograd.Function):
    """Compressor (mode 0) / expander (mode 1). Controls enter as separate tensors with bs elements
    each, in the reference's order: threshold_db, ratio, attack_ms, release_ms, knee_db,
    makeup_gain_db (functional.py:275-286); release_ms is unused, as in the reference."""

    @staticmethod
    def forward(ctx, x, mode, sample_rate, eps, lookahead, threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db):
        _lib.require_device(x, "x")
        ctx.meta = (x.dtype, [(c.dtype, c.shape) for c in (threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db)])
        ctx.empty = x.numel() == 0
        if ctx.empty:
            return torch.empty_like(x)
        with torch.cuda.device(x.device):
            ctls = (threshold_db, ratio, attack_ms, knee_db, makeup_gain_db)
            ctl = torch.stack([c.detach().reshape(-1).to(device=x.device, dtype=torch.float32) for c in ctls], dim=1).contiguous()
            need = any(ctx.needs_input_grad)
            y, saved, cfg = _dyn_forward(x, mode, sample_rate, eps, lookahead, ctl, need)
            if need:
                ctx.save_for_backward(*saved)
                ctx.cfg = cfg
        return y.to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xd, cm = ctx.meta
        if ctx.empty:
            return (torch.empty_like(gy), None, None, None, None) + tuple(torch.zeros(shape, dtype=dt, device=gy.device) for dt, shape in cm)
        x32 = ctx.saved_tensors[0]
        with torch.cuda.device(x32.device):
            gx, gctl = _dyn_backward(ctx.saved_tensors, ctx.cfg, gy)
        g = gctl.t().contiguous()      # rows: threshold, ratio, attack, knee, makeup
        rows = {0: g[0], 1: g[1], 2: g[2], 4: g[3], 5: g[4]}
        outs = []
        for i, ((dt, shape), need) in enumerate(zip(cm, ctx.needs_input_grad[5:])):
            if not need:
                outs.append(None)
            elif i == 3:               # release_ms: no path to the output (functional.py:340,343-344)
                outs.append(torch.zeros(shape, dtype=dt, device=x32.device))
            else:
                outs.append(rows[i].reshape(shape).to(dt))
        return (gx.to(xd) if ctx.needs_input_grad[0] else None, None, None, None, None) + tuple(outs)


class DynamicsMatrixFunction(torch.autograd.Function):
    """The same kernels on the six controls as one (bs, 6) matrix, columns in the reference's order (threshold_db, ratio, attack_ms,
    release_ms, knee_db, makeup_gain_db) - what Processor.process_normalized has after de-normalising (modules.py:159-187): one tensor in,
    one gradient matrix out (zero column for release_ms), no per-control slicing and stacking."""

    @staticmethod
    def forward(ctx, x, mode, sample_rate, eps, lookahead, controls):
        _lib.require_device(x, "x")
        _lib.require_same_device(x, controls=controls)
        _require_rows(x, controls, 6, "controls")
        ctx.meta = (x.dtype, controls.dtype, controls.shape)
        ctx.empty = x.numel() == 0
        if ctx.empty:
            return torch.empty_like(x)
        with torch.cuda.device(x.device):
            c32 = controls.detach().to(torch.float32)
            ctl = torch.cat([c32[:, :3], c32[:, 4:]], dim=1)         # (no index tensor: a list index is a host -> device copy per call)
            need = any(ctx.needs_input_grad)
            y, saved, cfg = _dyn_forward(x, mode, sample_rate, eps, lookahead, ctl, need)
            if need:
                ctx.save_for_backward(*saved)
                ctx.cfg = cfg
        return y.to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xd, cd, cshape = ctx.meta
        if ctx.empty:
            return torch.empty_like(gy), None, None, None, None, torch.zeros(cshape, dtype=cd, device=gy.device)
        x32 = ctx.saved_tensors[0]
        with torch.cuda.device(x32.device):
            gx, gctl = _dyn_backward(ctx.saved_tensors, ctx.cfg, gy)
            g6 = None
            if ctx.needs_input_grad[5]:
                g6 = torch.cat([gctl[:, :3], torch.zeros_like(gctl[:, :1]), gctl[:, 3:]], dim=1).to(cd)      # release_ms: zero column
        return gx.to(xd) if ctx.needs_input_grad[0] else None, None, None, None, None, g6


class DynamicsCtlFunction(torch.autograd.Function):
    """The dynamics kernels on the (bs, 5) fp32 control rows they read, [threshold_db, ratio, attack_ms, knee_db, makeup_gain_db], with the
    gradient returned in the same layout (the chain's fused control op, ChainControlsFunction, produces and consumes it)."""

    @staticmethod
    def forward(ctx, x, mode, sample_rate, eps, lookahead, ctl):
        _lib.require_device(x, "x")
        _lib.require_same_device(x, ctl=ctl)
        _require_rows(x, ctl, 5, "ctl")
        ctx.xdtype = x.dtype
        ctx.empty = x.numel() == 0
        if ctx.empty:
            return torch.empty_like(x)
        with torch.cuda.device(x.device):
            need = any(ctx.needs_input_grad)
            y, saved, cfg = _dyn_forward(x, mode, sample_rate, eps, lookahead, _f32c(ctl), need)
            if need:
                ctx.save_for_backward(*saved)
                ctx.cfg = cfg
        return y.to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        if ctx.empty:
            return torch.empty_like(gy), None, None, None, None, torch.zeros(gy.shape[0], 5, dtype=torch.float32, device=gy.device)
        with torch.cuda.device(gy.device):
            gx, gctl = _dyn_backward(ctx.saved_tensors, ctx.cfg, gy)
        return gx.to(ctx.xdtype) if ctx.needs_input_grad[0] else None, None, None, None, None, gctl


class ChainControlsFunction(torch.autograd.Function):
    """De-normalisation of the compressor's (bs, 6), the reverb's (bs, 25) and the gain's (bs, 1) normalised parameters of the reference's
    effect chain in one launch (dasp_chain_controls), in the layouts the kernels read: ctl (bs, 5) with the chain's final gain folded into
    the make-up gain, band gains (bs, 12), band decays (bs, 12), mix (bs); backward: one launch back to the three parameter tensors.
    lo, span: ctypes float[32] (compressor 0-5, reverb 6-30, gain 31)."""

    @staticmethod
    def forward(ctx, comp_pn, reverb_pn, gain_pn, lo, span, range_flag=None):
        _lib.require_device(comp_pn, "comp_params")
        _lib.require_same_device(comp_pn, reverb_params=reverb_pn, gain_params=gain_pn)
        B = comp_pn.shape[0]
        dev = comp_pn.device
        ctx.span, ctx.B = span, B
        ctx.dtypes = (comp_pn.dtype, reverb_pn.dtype, gain_pn.dtype)
        with torch.cuda.device(dev):
            buf = torch.empty(B, 30, dtype=torch.float32, device=dev)          # one allocation: ctl | gains | decays | mix
            flat = buf.view(-1)
            ctl, gains, decays, mix = flat[:5 * B].view(B, 5), flat[5 * B:17 * B].view(B, 12), flat[17 * B:29 * B].view(B, 12), flat[29 * B:]
            if B:
                call("dasp_chain_controls", ptr(_f32c(comp_pn)), ptr(_f32c(reverb_pn)), ptr(_f32c(gain_pn)), lo, span, ptr(ctl), ptr(gains),
                     ptr(decays), ptr(mix), ptr(range_flag), B, stream())
        return ctl, gains, decays, mix

    @staticmethod
    @once_differentiable
    def backward(ctx, gctl, ggain, gdecay, gmix):
        B = ctx.B
        dev = next(g for g in (gctl, ggain, gdecay, gmix) if g is not None).device
        with torch.cuda.device(dev):
            z = lambda g, *shape: _f32c(g) if g is not None else torch.zeros(*shape, dtype=torch.float32, device=dev)
            gc = torch.empty(B, 6, dtype=torch.float32, device=dev)
            gr = torch.empty(B, 25, dtype=torch.float32, device=dev)
            gg = torch.empty(B, 1, dtype=torch.float32, device=dev)
            if B:
                call("dasp_chain_controls_backward", ptr(z(gctl, B, 5)), ptr(z(ggain, B, 12)), ptr(z(gdecay, B, 12)), ptr(z(gmix, B)), ctx.span,
                     ptr(gc), ptr(gr), ptr(gg), B, stream())
        cd, rd, gd = ctx.dtypes
        return gc.to(cd), gr.to(rd), gg.to(gd), None, None, None


def chain_eq_compressor_forward(x, eq_pn, types, lo, span, sample_rate, ctl, mode=0, eps=1e-8, range_flag=None):
    """y = compressor(parametric_eq(x)) in one pass over x (dasp_chain_forward, csrc/chainfwd.hip): the EQ designed from its normalised
    (Bp, 18) parameter tensor (dasp_peq_prepare_norm: de-normalisation + RBJ design on the device), the compressor on its (B, 5) control rows
    [threshold_db, ratio, attack_ms, knee_db, makeup_gain_db]. Forward only - no autograd node, nothing saved: the reference's target
    synthesis (examples/style_transfer.py:293-299 runs the chain under no_grad every step). Refuses tensors that require a gradient."""
    _lib.require_device(x, "x")
    _lib.require_same_device(x, eq_params=eq_pn, ctl=ctl)
    from .ops64 import require_fp32_ok
    require_fp32_ok(x, "chain_eq_compressor_forward")       # fp32 kernel: float64 input is refused, not rounded behind the caller's back
    if torch.is_grad_enabled() and (x.requires_grad or eq_pn.requires_grad or ctl.requires_grad):
        raise RuntimeError("chain_eq_compressor_forward is forward-only: call it under torch.no_grad() or on detached tensors")
    _require_rows(x, ctl, 5, "ctl")
    L = _lib.lib()
    S = len(types)
    B, C, N = x.shape
    if eq_pn.dim() != 2 or eq_pn.shape[1] != 3 * S or eq_pn.shape[0] not in (1, B):
        raise RuntimeError(f"EQ parameters must be ({B} or 1, {3 * S}), got {tuple(eq_pn.shape)}")
    if x.numel() == 0:
        return torch.empty_like(x)
    dev = x.device
    with torch.cuda.device(dev):
        x32, pn32, c32 = _f32c(x), _f32c(eq_pn), _f32c(ctl)
        Bp = pn32.shape[0]
        tseg = 0 if not config.plan.chain_segment else int(config.plan.chain_segment_tiles or L.dasp_chain_segment_tiles(B, N))
        n_tab = _round64(Bp * L.dasp_sos_table_floats(S))
        n_seg = _round64(L.dasp_chain_seg_floats(B, C, N, S, tseg)) if tseg else 0
        f32 = torch.empty(n_tab + n_seg, dtype=torch.float32, device=dev)
        n_dt = Bp * L.dasp_sos_dtab_doubles(S)
        f64 = torch.empty(n_dt + (Bp * L.dasp_sos_segtab_doubles(S) if tseg else 0), dtype=torch.float64, device=dev)
        tab, segbuf = f32[:n_tab], (f32[n_tab:] if tseg else None)
        dtab, segtab = f64[:n_dt], (f64[n_dt:] if tseg else None)
        y = torch.empty_like(x32)
        call("dasp_peq_prepare_norm_seg", ptr(pn32), Bp, S, (ctypes.c_int * S)(*types), float(sample_rate), (ctypes.c_double * (3 * S))(*lo),
             (ctypes.c_double * (3 * S))(*span), ptr(range_flag), ptr(tab), ptr(dtab), tseg, ptr(segtab), stream())      # design (+ segment matrices; range_flag: see parametric_eq_norm)
        call("dasp_chain_forward", ptr(tab), Bp, ptr(x32), ptr(c32), ptr(y), B, C, N, S, int(mode), float(sample_rate), float(eps), tseg,
             ptr(segtab), ptr(segbuf), stream())
    return y.to(x.dtype)


def _cbuf(n, device):
    """n complex64 elements as a float32 buffer (the C ABI takes void*)."""
    return torch.empty(2 * n, dtype=torch.float32, device=device)


class _StereoFunction(torch.autograd.Function):
    """Shared plumbing of stereo_widener / stereo_panner / stereo_bus: y = f(x, ctl) with a small per-item / per-track control."""
    OP = None       # 0 widener, 1 panner, 2 bus
    FWD = BWD = None

    @classmethod
    def _dims(cls, x):
        if cls.OP == 0:
            B, _, N = x.shape
            return B, 1, N, (B, 2, N)
        if cls.OP == 1:
            B, T, N = x.shape
            return B, T, N, (B, 2, T, N)
        B, _, T, N = x.shape
        return B, T, N, (B, 2, N)

    @classmethod
    def _run(cls, ctx, x, ctl):
        _lib.require_device(x, "x")
        B, T, N, oshape = cls._dims(x)
        ctx.meta = (x.dtype, ctl.dtype, ctl.shape, B, T, N)
        ctx.empty = x.numel() == 0
        if ctx.empty:
            ctx.xshape = x.shape
            return torch.empty(oshape, dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            x32 = _f32c(x)
            c32 = ctl.detach().reshape(-1).to(device=x.device, dtype=torch.float32).contiguous()
            y = torch.empty(oshape, dtype=torch.float32, device=x.device)
            dims = (B, N) if cls.OP == 0 else (B, T, N)
            call(cls.FWD, ptr(x32), ptr(c32), ptr(y), *dims, stream())
            ctx.save_for_backward(x32, c32)
        return y.to(x.dtype)

    @classmethod
    def _grad(cls, ctx, gy):
        xd, cd, cshape, B, T, N = ctx.meta
        if ctx.empty:
            return torch.empty(ctx.xshape, dtype=xd, device=gy.device), torch.zeros(cshape, dtype=cd, device=gy.device)
        L = _lib.lib()
        x32, c32 = ctx.saved_tensors
        with torch.cuda.device(x32.device):
            gx = torch.empty_like(x32)
            gctl = torch.empty_like(c32)
            partials = torch.empty(L.dasp_stereo_partial_floats(cls.OP, B, T, N), dtype=torch.float32, device=x32.device)
            dims = (B, N) if cls.OP == 0 else (B, T, N)
            call(cls.BWD, ptr(x32), ptr(c32), ptr(_f32c(gy)), ptr(gx), ptr(gctl), ptr(partials), *dims, stream())
        return gx.to(xd), gctl.reshape(cshape).to(cd)


class WidenerFunction(_StereoFunction):
    """(L, R) -> (L + k R, k L + R), k = 1 - 2 width (functional.py:580-605)."""
    OP, FWD, BWD = 0, "dasp_widener_forward", "dasp_widener_backward"

    @staticmethod
    def forward(ctx, x, width):
        return WidenerFunction._run(ctx, x, width)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        return WidenerFunction._grad(ctx, gy)


class PannerFunction(_StereoFunction):
    """(B, T, N) mono tracks -> (B, 2, T, N) with the pan law of functional.py:608-636."""
    OP, FWD, BWD = 1, "dasp_panner_forward", "dasp_panner_backward"

    @staticmethod
    def forward(ctx, x, pan):
        return PannerFunction._run(ctx, x, pan)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        return PannerFunction._grad(ctx, gy)


class BusFunction(_StereoFunction):
    """(B, 2, T, N) -> (B, 2, N): sum of the tracks weighted by 10^(send_db / 20) (functional.py:32-62)."""
    OP, FWD, BWD = 2, "dasp_bus_forward", "dasp_bus_backward"

    @staticmethod
    def forward(ctx, x, send_db):
        return BusFunction._run(ctx, x, send_db)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        return BusFunction._grad(ctx, gy)


_FSPEC_CACHE = {}


def _filter_spectrum(filters, nb, taps, n_complex, dev):
    """Spectra of the filterbank taps (dasp_reverb_filter_spectrum). They depend only on the taps, so the result is kept
    for as long as the caller keeps passing the same (unmodified) filters tensor on the same stream -- functional.py
    holds one device copy of the bank per (taps, sample_rate, device)."""
    capturing = torch.cuda.is_current_stream_capturing()   # memory allocated inside a HIP-graph capture belongs to that graph
    key = (id(filters), filters._version, int(n_complex), int(torch.cuda.current_stream().cuda_stream))
    hit = None if capturing else _FSPEC_CACHE.get(key)
    if hit is not None and hit[0] is filters:
        return hit[1]
    Fspec = _cbuf(n_complex, dev)
    call("dasp_reverb_filter_spectrum", ptr(_f32c(filters)), nb, taps, ptr(Fspec), stream())
    if not capturing and filters.is_cuda and filters.dtype == torch.float32 and filters.is_contiguous() and not filters.requires_grad:
        if len(_FSPEC_CACHE) >= 8:
            _FSPEC_CACHE.clear()
        _FSPEC_CACHE[key] = (filters, Fspec)
    return Fspec


def _seed_offset(t, dev):
    """The optional per-replay seed offset: a 1-element int64 tensor on the op's device (read by the kernels when they run)."""
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype is torch.int64 and t.numel() == 1 and t.device == dev):
        raise ValueError("noise_seed_offset must be a 1-element int64 tensor on x's device")
    return t


def reverb_noise(seed, B, nb, row_len, device, seed_offset=None):
    """The white-noise stream the filter-bank kernels generate for `seed` (dasp_reverb_forward_rng), written out in the reference's layout
    (2B, nb, row_len) (dasp_pytorch/functional.py:548). A test / inspection hook: the product never materialises it."""
    dev = torch.device(device)
    out = torch.empty(2 * B, nb, row_len, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        call("dasp_reverb_noise", ctypes.c_ulonglong(int(seed) & 0xFFFFFFFFFFFFFFFF), ptr(_seed_offset(seed_offset, dev)), ptr(out), B, nb, row_len, stream())
    return out


class ReverbFunction(torch.autograd.Function):
    """noise_shaped_reverberation core: x (B,2,N) or mono (B,1,N) (the reference duplicates a mono input to stereo, functional.py:493-495; here
    both output channels read the one row - no copy; the backward adds the two channels' input gradients), output (B,2,N);
    noise (2B,nb,L+taps-1) or None, filters (nb,taps), gains/decays (B,nb), mix (B).
    `noise` and `filters` are constants of the op (the reference draws the noise inside the function, functional.py:548, and designs the
    filters with SciPy): asking for their gradient raises instead of silently returning None. noise = None: the noise is generated inside
    the filter-bank kernels from the integer `seed` (csrc/reverb.hip, counter-based: forward and backward recompute the same stream) and
    never exists in memory."""

    @staticmethod
    def forward(ctx, x, noise, filters, gains, decays, mix, L_ir, seed=None, seed_offset=None, decay_bound=0.0):
        _lib.require_device(x, "x")
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise RuntimeError("noise_shaped_reverberation: `noise` and `filters` are not differentiable inputs (detach them)")
        if noise is None and seed is None:
            raise ValueError("ReverbFunction: either a noise tensor or a seed")
        Lb = _lib.lib()
        B, C, N = x.shape
        nb, taps = filters.shape
        dev = x.device
        ctx.meta = (x.dtype, gains.dtype, gains.shape, decays.dtype, decays.shape, mix.dtype, mix.shape)
        ctx.empty = x.numel() == 0
        ctx.xC = C
        if C not in (1, 2):
            raise RuntimeError(f"noise_shaped_reverberation takes mono or stereo input, got {C} channels")
        if ctx.empty:
            return torch.empty(B, 2, N, dtype=x.dtype, device=dev)
        with torch.cuda.device(dev):
            sizes = (ctypes.c_long * 14)()
            check(Lb.dasp_reverb_sizes(B, N, L_ir, taps, nb, sizes), "dasp_reverb_sizes")
            if tuple(gains.shape) != (B, nb) or tuple(decays.shape) != (B, nb) or mix.numel() != B:
                # the kernels index gains[b * nb + band]: a (k, nb) stack with k != bs must not be reshaped into (bs, ...) silently
                # (the reference's torch.stack(...).view(bs, 12) raises for it, functional.py:498-544)
                raise RuntimeError(f"shape '[{B}, {nb}]' is invalid for band gains / decays of shapes {tuple(gains.shape)} / "
                                   f"{tuple(decays.shape)} and mix with {mix.numel()} values")
            x32 = _f32c(x)
            n32 = None
            if noise is not None:
                n32 = _f32c(noise)
                if n32.numel() != 2 * B * nb * (L_ir + taps - 1):
                    raise RuntimeError(f"noise must hold (2 * {B}, {nb}, {L_ir + taps - 1}) values, got {tuple(noise.shape)}")
            useed = ctypes.c_ulonglong(int(seed) & 0xFFFFFFFFFFFFFFFF) if noise is None else None
            soff = _seed_offset(seed_offset, dev) if noise is None else None
            g32, d32, m32 = (_f32c(t.reshape(B, -1)) for t in (gains, decays, mix))
            Fspec = _filter_spectrum(filters, nb, taps, sizes[4], dev)
            y = torch.empty(B, 2, N, dtype=torch.float32, device=dev)
            need_grad = any(ctx.needs_input_grad)
            # kept for the backward pass: the column transforms of x (A), the impulse responses (ir) and their spectra (H: one complex frame
            # per item); everything else is scratch
            A = _cbuf(sizes[6], dev) if need_grad else None
            W2 = None if need_grad else _cbuf(sizes[12], dev)
            W, H, Ah = _cbuf(sizes[12], dev), _cbuf(sizes[7], dev), _cbuf(sizes[13], dev)
            ir = torch.empty(sizes[8], dtype=torch.float32, device=dev)
            if noise is None:
                call("dasp_reverb_forward_rng", ptr(x32), useed, ptr(soff), ptr(Fspec), ptr(g32), ptr(d32), ptr(m32), ptr(y), ptr(A), ptr(H),
                     ptr(W), ptr(W2), ptr(Ah), ptr(ir), B, C, N, L_ir, taps, nb, float(decay_bound), stream())
            else:
                call("dasp_reverb_forward", ptr(x32), ptr(n32), ptr(Fspec), ptr(g32), ptr(d32), ptr(m32), ptr(y), ptr(A), ptr(H),
                     ptr(W), ptr(W2), ptr(Ah), ptr(ir), B, C, N, L_ir, taps, nb, float(decay_bound), stream())
            if need_grad:
                # the seed-offset word is read again by the backward kernels when they run: it is saved WITH the tensors, so that an in-place
                # bump between this forward and its backward (which would regenerate a different noise stream) trips autograd's
                # version check instead of giving silently wrong gain / decay / mix gradients (bump it after backward, or per replay)
                ctx.save_for_backward(ir, n32 if n32 is not None else x32.new_empty(0), Fspec, g32, d32, m32, A, H,
                                      soff if soff is not None else torch.empty(0, dtype=torch.int64, device=dev))
                ctx.cfg = (B, C, N, L_ir, taps, nb, [int(v) for v in sizes], useed, soff is not None, float(decay_bound))
        return y.to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xd, gd, gs, dd, ds, md, ms = ctx.meta
        if ctx.empty:
            z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=gy.device)
            return torch.empty(gy.shape[0], ctx.xC, gy.shape[2], dtype=xd, device=gy.device), None, None, z(gs, gd), z(ds, dd), z(ms, md), None, None, None, None
        ir, n32, Fspec, g32, d32, m32, A, H, soff = ctx.saved_tensors
        B, C, N, L_ir, taps, nb, sizes, useed, has_soff, dbound = ctx.cfg
        soff = soff if has_soff else None
        dev = ir.device
        with torch.cuda.device(dev):
            gx = torch.empty(B, 2, N, dtype=torch.float32, device=dev)
            ggain = torch.empty(B, nb, dtype=torch.float32, device=dev)
            gdecay = torch.empty(B, nb, dtype=torch.float32, device=dev)
            gmix = torch.empty(B, dtype=torch.float32, device=dev)
            Ag, W = _cbuf(sizes[12], dev), _cbuf(sizes[12], dev)
            P = _cbuf(sizes[13], dev)
            gir = torch.empty(sizes[8], dtype=torch.float32, device=dev)
            part = torch.empty(sizes[11], dtype=torch.float32, device=dev)
            mix_part = torch.empty(sizes[10], dtype=torch.float32, device=dev)
            tail = (ptr(Fspec), ptr(g32), ptr(d32), ptr(m32), ptr(A), ptr(H), ptr(gx), ptr(ggain), ptr(gdecay), ptr(gmix), ptr(Ag), ptr(W), ptr(P),
                    ptr(gir), ptr(part), ptr(mix_part), B, C, N, L_ir, taps, nb, dbound, stream())
            if useed is not None:
                call("dasp_reverb_backward_rng", ptr(ir), ptr(_f32c(gy)), useed, ptr(soff), *tail)
            else:
                call("dasp_reverb_backward", ptr(ir), ptr(_f32c(gy)), ptr(n32), *tail)
        if C == 1:
            gx = gx.sum(1, keepdim=True)           # the adjoint of the mono -> stereo duplication
        return gx.to(xd), None, None, ggain.reshape(gs).to(gd), gdecay.reshape(ds).to(dd), gmix.reshape(ms).to(md), None, None, None, None
