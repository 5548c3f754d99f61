"""float64 path: torch.autograd.Function wrappers over the double-precision entry points of the C ABI (csrc/ref64.hip).

The reference follows the dtype of its input (`.type_as(x)`, dasp_pytorch/signal.py:113,119, functional.py:211), so a float64 `x` means
float64 arithmetic. The fp32 kernels have no double instantiation; these plain sequential kernels are the genuine fp64 path for the
recurrences (parametric_eq / sosfilt_via_fsm / lfilter_via_fsm, compressor / expander) and the elementwise effects (gain, distortion):
right for validation and torch.autograd.gradcheck, not tuned for throughput. The FFT-based and stereo ops compute in fp32 only and
refuse float64 input unless config.plan.fp64_as_fp32 asks for the old cast-compute-cast behaviour (`require_fp32_ok`).
"""

import torch
from torch.autograd.function import once_differentiable

from . import _lib, config
from ._lib import call, ptr, stream


def is_f64(x):
    return isinstance(x, torch.Tensor) and x.dtype is torch.float64


def require_fp32_ok(x, what):
    """Ops without a double-precision path: float64 input raises instead of being rounded to fp32 behind the caller's back."""
    if is_f64(x) and not config.plan.fp64_as_fp32:
        raise _lib.DaspHipError(
            f"{what}: float64 input, but this op computes in float32 only. Cast the input (`x.float()`), or set dasp_pytorch_amd.config.plan.fp64_as_fp32 = True to have "
            "it cast, computed in fp32 and cast back (the result then has fp32 accuracy in a float64 tensor).")


def _d(t, dev):
    return t.detach().to(device=dev, dtype=torch.float64).contiguous()


class SosFilt64Function(torch.autograd.Function):
    """y = cascade of S biquads `sos` (Bs, S, 6), any a0, applied to x (B, C, N), all in float64."""

    @staticmethod
    def forward(ctx, sos, x):
        _lib.require_device(x, "x")
        _lib.require_same_device(x, sos=sos)
        Bs, S, _ = sos.shape
        B, C, N = x.shape
        dev = x.device
        ctx.meta = (sos.dtype, sos.shape, x.shape)
        if x.numel() == 0:
            ctx.empty = True
            return torch.empty_like(x)
        ctx.empty = False
        with torch.cuda.device(dev):
            s64, x64 = _d(sos, dev), _d(x, dev)
            c5 = torch.empty(Bs, S, 5, dtype=torch.float64, device=dev)
            call("dasp_sos64_normalize", ptr(s64), Bs, S, ptr(c5), stream())
            y = torch.empty_like(x64)
            wsave = torch.empty(B * C * S * N, dtype=torch.float64, device=dev) if ctx.needs_input_grad[0] else None
            call("dasp_sos64_forward", ptr(c5), Bs, ptr(x64), ptr(y), ptr(wsave), B, C, N, S, stream())
            ctx.save_for_backward(s64, c5, wsave if wsave is not None else torch.empty(0, device=dev))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        sdt, sshape, xshape = ctx.meta
        if ctx.empty:
            return torch.zeros(sshape, dtype=sdt, device=gy.device), torch.empty(xshape, dtype=torch.float64, device=gy.device)
        s64, c5, wsave = ctx.saved_tensors
        Bs, S, _ = sshape
        B, C, N = xshape
        dev = gy.device
        need_c = ctx.needs_input_grad[0]
        with torch.cuda.device(dev):
            gx = torch.empty(xshape, dtype=torch.float64, device=dev)
            gc5 = torch.empty(Bs, S, 5, dtype=torch.float64, device=dev) if need_c else None
            call("dasp_sos64_backward", ptr(c5), Bs, ptr(_d(gy, dev)), ptr(wsave if need_c else None), ptr(gx), ptr(gc5), B, C, N, S, stream())
            gsos = None
            if need_c:
                gsos = torch.empty(Bs, S, 6, dtype=torch.float64, device=dev)
                call("dasp_sos64_grads", ptr(c5), ptr(s64), ptr(gc5), None, Bs, S, 0, ptr(gsos), stream())
                gsos = gsos.to(sdt)
        return gsos, (gx if ctx.needs_input_grad[1] else None)


class LFilterFunction(torch.autograd.Function):
    """lfilter_via_fsm for K = 4 .. 16 coefficients (reference: signal.py:95-133): x (B, 1, N) float32 or float64, bn / an (Bs, K) with
    an[:, 0] == 1 (the caller normalises by a0 in torch, so that step differentiates itself), Bs = B or 1. Exact recurrence in double
    arithmetic (csrc/lfilter.hip); differentiable w.r.t. x, bn and an[:, 1:]."""

    @staticmethod
    def forward(ctx, x, bn, an):
        _lib.require_device(x, "x")
        _lib.require_same_device(x, b=bn, a=an)
        B, C, N = x.shape
        Bs, K = bn.shape
        dev = x.device
        ctx.meta = (x.dtype, x.shape, bn.dtype, an.dtype, bn.shape)
        ctx.empty = x.numel() == 0
        if ctx.empty:
            return torch.empty_like(x)
        f64 = x.dtype == torch.float64
        with torch.cuda.device(dev):
            xc = x.detach().to(torch.float64 if f64 else torch.float32).contiguous()
            b64, a64 = _d(bn, dev), _d(an, dev)
            y = torch.empty_like(xc)
            need = any(ctx.needs_input_grad)
            wsave = torch.empty(N * B * C, dtype=torch.float64, device=dev) if need else None
            # the chunk length is resolved ONCE per filter operation (developer override config.plan.lfilter_chunk, tests: chunk boundaries at odd
            # places) and handed to the size query, the forward and - through ctx - the backward call: they must cut time the same way
            chunk = ctx.chunk = max(int(config.plan.lfilter_chunk or 0), 0)
            nwork = _lib.lib().dasp_lfilter_work_doubles(B * C, N, K, chunk)
            work = torch.empty(nwork, dtype=torch.float64, device=dev) if nwork > 0 else None
            call("dasp_lfilter_forward", ptr(xc), ptr(b64), ptr(a64), Bs, ptr(y), ptr(wsave), ptr(work), nwork, B * C, N, K, int(f64), chunk, stream())
            if need:
                ctx.save_for_backward(b64, a64, wsave)
        return y.to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xdt, xshape, bdt, adt, bshape = ctx.meta
        if ctx.empty:
            return torch.empty(xshape, dtype=xdt, device=gy.device), torch.zeros(bshape, dtype=bdt, device=gy.device), torch.zeros(bshape, dtype=adt, device=gy.device)
        b64, a64, wsave = ctx.saved_tensors
        B, C, N = xshape
        Bs, K = bshape
        dev = gy.device
        f64 = xdt == torch.float64
        with torch.cuda.device(dev):
            g = gy.detach().to(torch.float64 if f64 else torch.float32).contiguous()
            gx = torch.empty_like(g) if ctx.needs_input_grad[0] else None
            gb = torch.empty(B * C, K, dtype=torch.float64, device=dev)
            ga = torch.empty(B * C, K, dtype=torch.float64, device=dev)
            nwork = _lib.lib().dasp_lfilter_work_doubles(B * C, N, K, ctx.chunk)
            work = torch.empty(nwork, dtype=torch.float64, device=dev) if nwork > 0 else None
            call("dasp_lfilter_backward", ptr(g), ptr(b64), ptr(a64), Bs, ptr(wsave), ptr(gx), ptr(gb), ptr(ga), ptr(work), nwork, B * C, N, K, int(f64), ctx.chunk, stream())
            if Bs == 1 and B * C > 1:
                gb, ga = gb.sum(0, keepdim=True), ga.sum(0, keepdim=True)
        return (gx.to(xdt) if gx is not None else None), gb.to(bdt), ga.to(adt)


class ParametricEQ64Function(torch.autograd.Function):
    """RBJ design in fp64 (dasp_biquad_design per section) + the fp64 cascade; `controls` as for ops.ParametricEQFunction."""

    @staticmethod
    def forward(ctx, x, sample_rate, types, *controls):
        _lib.require_device(x, "x")
        S = len(types)
        B, C, N = x.shape
        dev = x.device
        ctx.ctl = [(c.dtype, c.shape) for c in controls]
        ctx.xshape = x.shape
        if x.numel() == 0:
            ctx.empty = True
            return torch.empty_like(x)
        ctx.empty = False
        with torch.cuda.device(dev):
            cols = [_d(c, dev).reshape(-1) for c in controls]
            Bp = cols[0].numel()
            if any(c.numel() != Bp for c in cols):
                raise ValueError("parametric_eq controls must all have the same number of elements")
            ba = torch.empty(S, Bp, 6, dtype=torch.float64, device=dev)
            jac = torch.empty(S, Bp, 15, dtype=torch.float64, device=dev)
            for k in range(S):
                call("dasp_biquad_design", ptr(cols[3 * k]), ptr(cols[3 * k + 1]), ptr(cols[3 * k + 2]), Bp, int(types[k]), float(sample_rate),
                     ptr(ba[k]), ptr(jac[k]), stream())
            c5 = torch.cat([ba[:, :, :3], ba[:, :, 4:]], dim=2).permute(1, 0, 2).contiguous()          # (Bp, S, 5): a re-layout, no arithmetic
            x64 = _d(x, dev)
            y = torch.empty_like(x64)
            need_c = any(ctx.needs_input_grad[3:])
            wsave = torch.empty(B * C * S * N, dtype=torch.float64, device=dev) if need_c else None
            call("dasp_sos64_forward", ptr(c5), Bp, ptr(x64), ptr(y), ptr(wsave), B, C, N, S, stream())
            ctx.save_for_backward(c5, jac.permute(1, 0, 2).contiguous(), wsave if need_c else torch.empty(0, device=dev))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        if ctx.empty:
            return (torch.empty(ctx.xshape, dtype=torch.float64, device=gy.device), None, None) + tuple(
                torch.zeros(shape, dtype=dt, device=gy.device) for dt, shape in ctx.ctl)
        c5, jac, wsave = ctx.saved_tensors
        Bp, S, _ = c5.shape
        B, C, N = ctx.xshape
        dev = gy.device
        need_c = any(ctx.needs_input_grad[3:])
        with torch.cuda.device(dev):
            gx = torch.empty(ctx.xshape, dtype=torch.float64, device=dev)
            gc5 = torch.empty(Bp, S, 5, dtype=torch.float64, device=dev) if need_c else None
            call("dasp_sos64_backward", ptr(c5), Bp, ptr(_d(gy, dev)), ptr(wsave if need_c else None), ptr(gx), ptr(gc5), B, C, N, S, stream())
            gcols = (None,) * len(ctx.ctl)
            if need_c:
                gp = torch.empty(Bp, S, 3, dtype=torch.float64, device=dev)
                call("dasp_sos64_grads", ptr(c5), None, ptr(gc5), ptr(jac), Bp, S, 1, ptr(gp), stream())
                flat = gp.reshape(Bp, 3 * S)
                gcols = tuple(flat[:, i].reshape(shape).to(dt) if need else None
                              for i, ((dt, shape), need) in enumerate(zip(ctx.ctl, ctx.needs_input_grad[3:])))
        return (gx if ctx.needs_input_grad[0] else None, None, None) + gcols


class Dynamics64Function(torch.autograd.Function):
    """compressor (mode 0) / expander (mode 1) in float64; arguments as ops.DynamicsFunction."""

    @staticmethod
    def forward(ctx, x, mode, sample_rate, eps, lookahead, threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db):
        _lib.require_device(x, "x")
        B, C, N = x.shape
        dev = x.device
        ctx.meta = [(c.dtype, c.shape) for c in (threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db)]
        ctx.xshape = x.shape
        if x.numel() == 0:
            ctx.empty = True
            return torch.empty_like(x)
        ctx.empty = False
        with torch.cuda.device(dev):
            ctl = torch.stack([_d(c, dev).reshape(-1) for c in (threshold_db, ratio, attack_ms, knee_db, makeup_gain_db)], dim=1).contiguous()
            x64 = _d(x, dev)
            y = torch.empty_like(x64)
            need = any(ctx.needs_input_grad)
            gsave = torch.empty(B, N, dtype=torch.float64, device=dev) if need else None
            call("dasp_dynamics64_forward", mode, ptr(x64), ptr(ctl), ptr(y), ptr(gsave), B, C, N, float(sample_rate), float(eps), int(lookahead), stream())
            if need:
                ctx.save_for_backward(x64, ctl, gsave)
                ctx.cfg = (mode, float(sample_rate), float(eps), int(lookahead))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        dev = gy.device
        if ctx.empty:
            return (torch.empty(ctx.xshape, dtype=torch.float64, device=dev), None, None, None, None) + tuple(
                torch.zeros(shape, dtype=dt, device=dev) for dt, shape in ctx.meta)
        x64, ctl, gsave = ctx.saved_tensors
        mode, sr, eps, look = ctx.cfg
        B, C, N = ctx.xshape
        with torch.cuda.device(dev):
            gx = torch.empty_like(x64)
            gctl = torch.empty(B, 5, dtype=torch.float64, device=dev)
            call("dasp_dynamics64_backward", mode, ptr(x64), ptr(ctl), ptr(_d(gy, dev)), ptr(gsave), ptr(gx), ptr(gctl), B, C, N, sr, eps, look, stream())
        rows = {0: gctl[:, 0], 1: gctl[:, 1], 2: gctl[:, 2], 4: gctl[:, 3], 5: gctl[:, 4]}
        outs = []
        for i, ((dt, shape), need) in enumerate(zip(ctx.meta, ctx.needs_input_grad[5:])):
            if not need:
                outs.append(None)
            elif i == 3:               # release_ms: no path to the output (functional.py:340,343-344)
                outs.append(torch.zeros(shape, dtype=dt, device=dev))
            else:
                outs.append(rows[i].reshape(shape).to(dt))
        return (gx if ctx.needs_input_grad[0] else None, None, None, None, None) + tuple(outs)


class Elementwise64Function(torch.autograd.Function):
    """op 0: gain (ctl: bs values), op 1: distortion (ctl: bs * chs values), float64."""

    @staticmethod
    def forward(ctx, x, ctl, op):
        _lib.require_device(x, "x")
        B, C, N = x.shape
        dev = x.device
        ctx.meta = (ctl.dtype, ctl.shape, x.shape, op)
        if x.numel() == 0:
            ctx.empty = True
            return torch.empty_like(x)
        ctx.empty = False
        with torch.cuda.device(dev):
            x64, c64 = _d(x, dev), _d(ctl, dev).reshape(-1)
            y = torch.empty_like(x64)
            call("dasp_ew64_forward", op, ptr(x64), ptr(c64), ptr(y), B, C, N, stream())
            ctx.save_for_backward(x64, c64)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        cdt, cshape, xshape, op = ctx.meta
        dev = gy.device
        if ctx.empty:
            return torch.empty(xshape, dtype=torch.float64, device=dev), torch.zeros(cshape, dtype=cdt, device=dev), None
        x64, c64 = ctx.saved_tensors
        B, C, N = xshape
        with torch.cuda.device(dev):
            gx = torch.empty_like(x64)
            gctl = torch.empty_like(c64)
            call("dasp_ew64_backward", op, ptr(x64), ptr(c64), ptr(_d(gy, dev)), ptr(gx), ptr(gctl), B, C, N, stream())
        return gx, gctl.reshape(cshape).to(cdt), None
