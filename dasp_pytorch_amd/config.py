"""Developer switches of the Python layer, as plain attributes of `plan`. Nothing in the package reads an environment variable for its
behaviour (rounds 2 - 5 had a dozen DASP_* switches; csrc/ lost its getenv calls in round 5, the Python layer in round 6) - the one
exception is DASP_HIP_LIB, the path of an alternative build of the kernel library for same-box A/B measurements (_lib.py), which has
to be known before anything is imported. Tests set these with `monkeypatch.setattr(config.plan, ...)` or `config.override(...)`.

    sos_segment / dyn_segment / chain_segment     False: never cut rows / items into segments (one workgroup per row / item)
    *_segment_tiles                               tiles per segment instead of the library planner's choice (None = the planner)
    torch_ops                                     False: the ctypes autograd binding (_ctypes_ops.py) instead of torch.ops.dasp.* (csrc/torch_ext)
    chain_fused_controls / chain_fused_forward    False: StyleTransferChain without its fused control launch / fused no-grad EQ + compressor pass
    chain_fused_grad                              the EQ -> compressor forward of the pass WITH gradients as one launch that saves for both
                                                  backward passes: None = from 384 rows (192 stereo items) on, where it wins, True = always, False = never
    fp64_as_fp32                                  True: ops without a double-precision path cast float64 input instead of raising
    lfilter_chunk                                 samples per chunk of time of csrc/lfilter.hip (0 = the library's plan)
    lookback                                      False: the segmented launches in their two-launch forms (pre-pass + pass: no workgroup ever
                                                  waits for another) instead of the one-launch look-back forms; kept in the kernel library
                                                  (dasp_plan_lookback), so it holds for both bindings"""
import contextlib


class _Plan:
    sos_segment = True
    sos_segment_tiles = None
    dyn_segment = True
    dyn_segment_tiles = None
    chain_segment = True
    chain_segment_tiles = None
    torch_ops = True
    chain_fused_controls = True
    chain_fused_forward = True
    chain_fused_grad = None
    fp64_as_fp32 = False
    lfilter_chunk = 0

    @property
    def lookback(self):
        from . import _lib
        return bool(_lib.lib().dasp_plan_lookback(-1))

    @lookback.setter
    def lookback(self, on):
        from . import _lib
        _lib.lib().dasp_plan_lookback(1 if on else 0)


plan = _Plan()


@contextlib.contextmanager
def override(**kw):
    """with config.override(dyn_segment=False): ...   (restores the previous values on exit)"""
    old = {k: getattr(plan, k) for k in kw}          # AttributeError on a name that is not a switch
    try:
        for k, v in kw.items():
            setattr(plan, k, v)
        yield plan
    finally:
        for k, v in old.items():
            setattr(plan, k, v)
