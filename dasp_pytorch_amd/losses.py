"""Multi-resolution STFT loss on the HIP kernels of csrc/stftloss.hip: the loss the reference's training loops put directly after
the effect chain (auraloss.freq.MultiResolutionSTFTLoss(), examples/style_transfer.py:341,363). Same defaults and call convention as
auraloss 0.4.0: `loss_fn(input, target)` with (bs, chs, seq_len) tensors, spectral convergence + log-magnitude L1 per resolution,
mean over the resolutions. Only `input` receives a gradient (the target is the reference signal at every call site)."""
import ctypes

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import call, ptr, stream


class _MRSTFTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, target, res, eps):
        _lib.require_device(inp, "input")
        _lib.require_device(target, "target")
        _lib.require_same_device(inp, target=target)
        from .ops64 import require_fp32_ok
        require_fp32_ok(inp, "MultiResolutionSTFTLoss")
        if inp.shape != target.shape:
            raise RuntimeError(f"input {tuple(inp.shape)} and target {tuple(target.shape)} must have the same shape")
        L = _lib.lib()
        N = inp.shape[-1]
        p32 = inp.detach().reshape(-1, N).to(torch.float32).contiguous()
        t32 = target.detach().reshape(-1, N).to(torch.float32).contiguous()
        rows = p32.shape[0]
        nres = len(res)
        arr = [(ctypes.c_int * nres)(*[int(r[i]) for r in res]) for i in range(3)]
        nfl = L.dasp_mrstft_partial_floats(rows, N, nres, *arr)
        if nfl < 0:
            raise _lib.DaspHipError("unsupported STFT resolutions (fft a power of two in 8..4096, win <= fft, fft / 2 < seq_len, <= 8 of them)")
        dev = inp.device
        with torch.cuda.device(dev):
            tw = _twiddles(dev)
            partials = torch.empty(nfl, dtype=torch.float32, device=dev)
            stats = torch.empty(4 * nres, dtype=torch.float32, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            call("dasp_mrstft_forward", ptr(p32), ptr(t32), ptr(tw), ptr(partials), ptr(stats), ptr(loss), rows, N, nres, *arr, float(eps), stream())
        ctx.save_for_backward(p32, t32, stats, tw)
        ctx.cfg = (rows, N, nres, arr, float(eps), inp.shape, inp.dtype)
        ctx.tdtype = target.dtype
        return loss.to(inp.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss):
        p32, t32, stats, tw = ctx.saved_tensors
        rows, N, nres, arr, eps, shape, dtype = ctx.cfg
        g = gt = None
        with torch.cuda.device(p32.device):
            gl = gloss.detach().reshape(1).to(torch.float32).contiguous()
            if ctx.needs_input_grad[0]:
                g = torch.empty_like(p32)
                call("dasp_mrstft_backward", ptr(p32), ptr(t32), ptr(tw), ptr(stats), ptr(gl), ptr(g), rows, N, nres, *arr, eps, stream())
                g = g.reshape(shape).to(dtype)
            if ctx.needs_input_grad[1]:      # auraloss differentiates both arguments (a consistency loss between two model outputs)
                gt = torch.empty_like(t32)
                call("dasp_mrstft_backward_target", ptr(p32), ptr(t32), ptr(tw), ptr(stats), ptr(gl), ptr(gt), rows, N, nres, *arr, eps, stream())
                gt = gt.reshape(shape).to(ctx.tdtype)
        return g, gt, None, None


_TW = {}


def _twiddles(device):
    """The 4096-entry twiddle table, one per (device, stream). Inside a HIP-graph capture the table is built fresh and not kept: memory
    allocated while capturing belongs to that graph's pool, and its fill kernel only runs on replay (the same rule as
    ops._filter_spectrum). Keyed by stream as well: the fill is ordered only against work on the stream it was launched on."""
    capturing = torch.cuda.is_current_stream_capturing()
    key = (device.type, device.index, int(torch.cuda.current_stream(device).cuda_stream))
    if not capturing and key in _TW:
        return _TW[key]
    tw = torch.empty(2 * 4096, dtype=torch.float32, device=device)
    call("dasp_mrstft_table", ptr(tw), stream())
    if not capturing:
        if len(_TW) >= 16:
            _TW.clear()
        _TW[key] = tw
    return tw


class MultiResolutionSTFTLoss(torch.nn.Module):
    """auraloss.freq.MultiResolutionSTFTLoss with its default weights (w_sc = w_log_mag = 1, w_lin_mag = w_phs = 0, hann window,
    L1 magnitude distance, mean reduction)."""

    def __init__(self, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240), eps: float = 1e-8):
        super().__init__()
        if not (len(fft_sizes) == len(hop_sizes) == len(win_lengths)):
            raise ValueError("fft_sizes, hop_sizes and win_lengths must have the same length")
        self.resolutions = tuple(zip(fft_sizes, hop_sizes, win_lengths))
        self.eps = eps

    def forward(self, input: torch.Tensor, target: torch.Tensor):
        return _MRSTFTFunction.apply(input, target, self.resolutions, self.eps)


class STFTLoss(MultiResolutionSTFTLoss):
    """auraloss.freq.STFTLoss with its default arguments (one resolution: fft 1024, hop 256, window 1024; w_sc = w_log_mag = 1, hann window,
    L1 magnitude distance, mean reduction) - the loss of the reference's examples/blind_estimation.py:141. One resolution of the same
    kernels (csrc/stftloss.hip)."""

    def __init__(self, fft_size: int = 1024, hop_size: int = 256, win_length: int = 1024, eps: float = 1e-8):
        super().__init__((fft_size,), (hop_size,), (win_length,), eps)


def mrstft_loss(input: torch.Tensor, target: torch.Tensor, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240),
                eps: float = 1e-8):
    return _MRSTFTFunction.apply(input, target, tuple(zip(fft_sizes, hop_sizes, win_lengths)), eps)
