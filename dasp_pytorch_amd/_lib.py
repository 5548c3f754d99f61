"""ctypes binding of libdasp_hip.so (C ABI: include/dasp_hip.h).

There is deliberately no CPU or PyTorch fallback: if the HIP library is missing, or a tensor is
not on a ROCm device, the call raises."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DASP_HIP_LIB") or os.path.join(_HERE, "csrc", "libdasp_hip.so")   # env override: A/B of builds
_lib = None

c_f = ctypes.c_void_p   # device pointers are passed as raw addresses
_i, _l, _d, _p = ctypes.c_int, ctypes.c_long, ctypes.c_double, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/dasp_hip.h one to one
SIGNATURES = {
    "dasp_abi_hash": (ctypes.c_ulonglong, []),
    "dasp_sos_supported_sections": (_i, [_i]),
    "dasp_sos_chunk": (_i, []),
    "dasp_sos_tile": (_i, []),
    "dasp_sos_bwd_waves": (_i, []),
    "dasp_sos_table_floats": (_l, [_i]),
    "dasp_sos_dtab_doubles": (_l, [_i]),
    "dasp_sos_num_tiles": (_l, [_l]),
    "dasp_sos_carry_floats": (_l, [_l, _l, _i]),
    "dasp_sos_partial_floats": (_l, [_l, _i]),
    "dasp_sos_prepare": (_i, [_p, _i, _i, _p, _p, _p]),
    "dasp_peq_prepare": (_i, [_p, _i, _i, ctypes.POINTER(ctypes.c_int), _d, _p, _p, _p]),
    "dasp_peq_prepare_rows": (_i, [ctypes.POINTER(ctypes.c_void_p), _i, _i, ctypes.POINTER(ctypes.c_int), _d, _p, _p, _p]),
    "dasp_sosfilt_forward": (_i, [_p, _i, _p, _p, _p, _i, _i, _l, _i, _p]),
    "dasp_sosfilt_backward": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _l, _i, _p]),
    "dasp_sos_grad_finalize": (_i, [_p, _i, _p, _i, _i, _i, _i, _p, _p]),
    "dasp_sosfilt_backward_grads": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _i, _l, _i, _p]),
    "dasp_sosfilt_backward_ex": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _l, _i, _p]),
    "dasp_sos_grad_finalize_ex": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p, _p]),
    "dasp_sosfilt_backward_grads_ex": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _i, _l, _i, _p]),
    "dasp_peq_forward": (_i, [ctypes.POINTER(ctypes.c_void_p), _i, _i, ctypes.POINTER(ctypes.c_int), _d, _p, _p, _p, _p, _p, _i, _i, _l, _l, _p, _p, _p]),
    "dasp_peq_forward_norm": (_i, [_p, _i, _i, ctypes.POINTER(ctypes.c_int), _d, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _p,
                                   _p, _p, _p, _p, _p, _i, _i, _l, _l, _p, _p, _p]),
    "dasp_peq_prepare_norm": (_i, [_p, _i, _i, ctypes.POINTER(ctypes.c_int), _d, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _p, _p, _p, _p]),
    "dasp_peq_prepare_norm_seg": (_i, [_p, _i, _i, ctypes.POINTER(ctypes.c_int), _d, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _p, _p, _p,
                                       _l, _p, _p]),
    "dasp_sos_segment_starts": (_i, [_p, _p, _i, _p, _p, _i, _i, _l, _i, _l, _p]),
    "dasp_chain_segment_tiles": (_l, [_l, _l]),
    "dasp_chain_seg_floats": (_l, [_l, _l, _l, _i, _l]),
    "dasp_chain_forward_saving": (_i, [_p] * 8 + [_i, _i, _l, _i, _i, _d, ctypes.c_float, _p]),
    "dasp_chain_forward": (_i, [_p, _i, _p, _p, _p, _i, _i, _l, _i, _i, _d, ctypes.c_float, _l, _p, _p, _p]),
    "dasp_peq_backward": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _i, _l, _i, _l, _p, _p, _p]),
    "dasp_sosfilt_backward_seg_ex": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _l, _i, _l, _p]),
    "dasp_biquad_design": (_i, [_p, _p, _p, _i, _i, _d, _p, _p, _p]),
    "dasp_biquad_backward": (_i, [_p, _p, _i, _p, _p]),
    "dasp_sos_segment_tiles": (_l, [_l, _l]),
    "dasp_sos_segments": (_l, [_l, _l]),
    "dasp_sos_segtab_doubles": (_l, [_i]),
    "dasp_sos_seg_floats": (_l, [_l, _l, _i, _l]),
    "dasp_sos_segment_prepare": (_i, [_p, _i, _i, _l, _p, _p]),
    "dasp_sosfilt_forward_seg": (_i, [_p, _p, _i, _p, _p, _p, _p, _i, _i, _l, _i, _l, _p]),
    "dasp_sosfilt_backward_seg": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _l, _i, _l, _p]),
    "dasp_sos_grad_finalize_seg": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p, _p]),
    "dasp_ew_partial_floats": (_l, [_l, _l]),
    "dasp_gain_forward": (_i, [_p, _p, _p, _i, _i, _l, _p]),
    "dasp_gain_backward": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _l, _p]),
    "dasp_distortion_forward": (_i, [_p, _p, _p, _i, _i, _l, _p]),
    "dasp_distortion_backward": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _l, _p]),
    "dasp_chain_controls": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "dasp_chain_controls_backward": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "dasp_distortion_sample_forward": (_i, [_p, _p, _p, _l, _p]),
    "dasp_distortion_sample_backward": (_i, [_p, _p, _p, _p, _p, _l, _p]),
    "dasp_dyn_num_tiles": (_l, [_l]),
    "dasp_dyn_carry_floats": (_l, [_l, _l]),
    "dasp_dyn_partial_floats": (_l, [_l]),
    "dasp_dyn_counters_reset": (_i, [_p, _i, _p]),
    "dasp_dynamics_forward": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _l, _d, ctypes.c_float, _i, _p]),
    "dasp_dynamics_backward": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _l, _d, ctypes.c_float, _i, _p]),
    "dasp_dyn_segment_tiles": (_l, [_l, _l]),
    "dasp_dyn_segments": (_l, [_l, _l]),
    "dasp_dynamics_forward_seg": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _l, _d, ctypes.c_float, _i, _l, _p, _p]),
    "dasp_dynamics_backward_seg": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _l, _d, ctypes.c_float, _i, _l, _p, _p]),
    "dasp_dynamics_forward_rows": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _l, _d, ctypes.c_float, _i, _l, _p, _p]),
    "dasp_dynamics_backward_rows": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _l, _d, ctypes.c_float, _i, _l, _p, _p]),
    "dasp_stereo_partial_floats": (_l, [_i, _l, _i, _l]),
    "dasp_widener_forward": (_i, [_p, _p, _p, _i, _l, _p]),
    "dasp_widener_backward": (_i, [_p] * 6 + [_i, _l, _p]),
    "dasp_panner_forward": (_i, [_p, _p, _p, _i, _i, _l, _p]),
    "dasp_panner_backward": (_i, [_p] * 6 + [_i, _i, _l, _p]),
    "dasp_bus_forward": (_i, [_p, _p, _p, _i, _i, _l, _p]),
    "dasp_bus_backward": (_i, [_p] * 6 + [_i, _i, _l, _p]),
    "dasp_mrstft_partial_floats": (_l, [_l, _i, _i, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "dasp_mrstft_table": (_i, [_p, _p]),
    "dasp_mrstft_forward": (_i, [_p] * 6 + [_i, _i, _i, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_float, _p]),
    "dasp_mrstft_backward": (_i, [_p] * 6 + [_i, _i, _i, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_float, _p]),
    "dasp_mrstft_backward_target": (_i, [_p] * 6 + [_i, _i, _i, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_float, _p]),
    "dasp_lfilter_work_doubles": (_l, [_i, _l, _i, _l]),
    "dasp_lfilter_forward": (_i, [_p, _p, _p, _i, _p, _p, _p, _l, _i, _l, _i, _i, _l, _p]),
    "dasp_lfilter_backward": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, _l, _i, _l, _i, _i, _l, _p]),
    "dasp_sos64_normalize": (_i, [_p, _i, _i, _p, _p]),
    "dasp_sos64_forward": (_i, [_p, _i, _p, _p, _p, _i, _i, _l, _i, _p]),
    "dasp_sos64_backward": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _l, _i, _p]),
    "dasp_sos64_grads": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "dasp_dynamics64_forward": (_i, [_i, _p, _p, _p, _p, _i, _i, _l, _d, _d, _i, _p]),
    "dasp_dynamics64_backward": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _l, _d, _d, _i, _p]),
    "dasp_ew64_forward": (_i, [_i, _p, _p, _p, _i, _i, _l, _p]),
    "dasp_ew64_backward": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _l, _p]),
    "dasp_reverb_plan": (_i, [_l, ctypes.c_float, _i]),
    "dasp_reverb_sizes": (_i, [_i, _l, _i, _i, _i, ctypes.POINTER(ctypes.c_long)]),
    "dasp_reverb_filter_spectrum": (_i, [_p, _i, _i, _p, _p]),
    "dasp_reverb_forward": (_i, [_p] * 13 + [_i, _i, _l, _i, _i, _i, ctypes.c_float, _p]),
    "dasp_reverb_backward": (_i, [_p] * 19 + [_i, _i, _l, _i, _i, _i, ctypes.c_float, _p]),
    "dasp_reverb_forward_rng": (_i, [_p, ctypes.c_ulonglong, _p] + [_p] * 11 + [_i, _i, _l, _i, _i, _i, ctypes.c_float, _p]),
    "dasp_reverb_backward_rng": (_i, [_p, _p, ctypes.c_ulonglong, _p] + [_p] * 16 + [_i, _i, _l, _i, _i, _i, ctypes.c_float, _p]),
    "dasp_reverb_noise": (_i, [ctypes.c_ulonglong, _p, _p, _i, _i, _l, _p]),
    "dasp_device_error": (_i, []),
    "dasp_device_error_clear": (None, []),
    "dasp_plan_lookback": (_i, [_i]),
    "dasp_test_lookback_timeout": (_i, [_i]),
    "dasp_test_lookback_stall": (_i, [_p, _p, _p]),
    "dasp_test_spin": (_i, [_i, _i, _d, _p]),
    "dasp_test_stream_with_cus": (_i, [_i, ctypes.POINTER(ctypes.c_void_p)]),
    "dasp_test_stream_destroy": (_i, [_p]),
    "dasp_mt_layout": (_i, [ctypes.POINTER(ctypes.c_int)]),
    "dasp_mt_max_values": (ctypes.c_longlong, []),
    "dasp_mt_scratch_words": (_l, [_i, ctypes.c_longlong]),
    "dasp_mt_randn": (_i, [_p, _i, _p, ctypes.c_longlong, _p, _p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_long), _p]),
}


class DaspHipError(RuntimeError):
    pass


def lib():
    """Load libdasp_hip.so once. Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DaspHipError(
                f"{LIB_PATH} not found: build it with `python -m dasp_pytorch_amd.csrc.build` "
                "(or __graft_entry__.build()). dasp_pytorch_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
        if torch.cuda.is_available():
            L.dasp_device_error()           # allocates the current device's error words now - not under some later stream capture
    return _lib



on_failure = []          # callables run when an entry point fails (_ctypes_ops.py: the cached completion counters may hold a count - drop them)


def check(status, what):
    if status != 0:
        for hook in on_failure:
            hook()
        kind = {-1: "invalid argument", -2: "unsupported configuration",
                -3: "DASP_ERR_DEVICE: a kernel of an EARLIER call on this device gave up waiting for a word from another workgroup or wave and wrote NaN "
                    f"(family bits {lib().dasp_device_error():#x}: 1 EQ forward, 2 EQ backward, 4 dynamics forward, 8 dynamics backward, 32 random stream); results since then "
                    "are not to be trusted. dasp_pytorch_amd.config.plan.lookback = False selects the two-launch forms; _lib.lib().dasp_device_error_clear() re-arms"
                }.get(status, f"hipError {status}")
        raise DaspHipError(f"{what} failed: {kind}")


class KernelTimers:
    """Optional per-entry-point HIP event timing (bench.py's roofline leg). Events are recorded on
    torch's current stream, which is the stream every kernel is launched on (see `stream()`)."""

    def __init__(self):
        self.enabled = False
        self.pending = {}
        self.every = 1        # record every n-th call of each entry point (events cost ~2 us each on the stream)
        self.count = {}

    def start(self, every=1):
        self.enabled, self.pending, self.every, self.count = True, {}, max(1, int(every)), {}

    def stop(self):
        """Synchronise and return {entry point: [ms per launch, ...]}."""
        self.enabled = False
        torch.cuda.synchronize()
        out = {k: [a.elapsed_time(b) for a, b in v] for k, v in self.pending.items()}
        self.pending = {}
        return out


timers = KernelTimers()


def call(name, *args):
    """Invoke one C-ABI entry point, raising DaspHipError on a non-zero status."""
    fn = getattr(lib(), name)
    sample = False
    if timers.enabled:
        n = timers.count.get(name, 0)
        timers.count[name] = n + 1
        sample = n % timers.every == 0
    if sample:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        status = fn(*args)
        e1.record()
        timers.pending.setdefault(name, []).append((e0, e1))
    else:
        status = fn(*args)
    check(status, name)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_same_device(x, **others):
    """Every tensor of a call lives on x's device (the kernels get raw pointers: a pointer of another GPU is a memory fault)."""
    for name, t in others.items():
        if isinstance(t, torch.Tensor) and t.device != x.device:
            raise DaspHipError(f"{name} is on {t.device} but x is on {x.device}: all tensors of a call must share one device")


def require_device(t, name="x"):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise DaspHipError(
            f"{name} is on {t.device}: dasp_pytorch_amd runs on MI355X (ROCm) devices only and has no CPU path")
