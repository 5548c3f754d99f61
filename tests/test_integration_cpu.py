"""The maintainer-side patch of INTEGRATION.md section 3, executed: the code block is taken from the document itself, run in front of
the reference's own dasp_pytorch/__init__.py (as the text says a maintainer would place it), and the resulting package is checked -
every Processor routes device tensors to the dasp_pytorch_amd functions and CPU tensors to the reference's own code.

Needs the reference checkout (/root/reference); skipped where it does not exist (the GPU box). Nothing is copied: the reference's
__init__.py is read and executed in place."""
import os
import re
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/dasp_pytorch"


def _stub_source():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## 3. Patch inside the reference"):md.index("## 4.")]
    return re.search(r"```python\n(.*?)```", sec, re.S).group(1)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_integration_stub_routes_processors(monkeypatch):
    if torch.version.hip is None:
        pytest.skip("the stub is conditional on a ROCm build of torch")
    import dasp_pytorch_amd.functional as amd_f
    import dasp_pytorch_amd.signal as amd_s
    seen = []
    for mod, names in ((amd_f, ("gain", "distortion", "parametric_eq", "compressor", "noise_shaped_reverberation", "stereo_widener",
                                "stereo_panner", "stereo_bus")), (amd_s, ("sosfilt_via_fsm", "lfilter_via_fsm", "biquad"))):
        for n in names:
            monkeypatch.setattr(mod, n, (lambda n: lambda *a, **k: seen.append(n) or ("amd", n))(n))
    saved = {k: v for k, v in sys.modules.items() if k == "dasp_pytorch" or k.startswith("dasp_pytorch.")}
    for k in saved:
        del sys.modules[k]
    try:
        pkg = types.ModuleType("dasp_pytorch")
        pkg.__path__ = [REF]
        pkg.__package__ = "dasp_pytorch"
        pkg.__file__ = os.path.join(REF, "__init__.py")
        sys.modules["dasp_pytorch"] = pkg
        exec(compile(_stub_source(), "INTEGRATION.md#3", "exec"), pkg.__dict__)                       # the maintainer's block first ...
        exec(compile(open(pkg.__file__).read(), pkg.__file__, "exec"), pkg.__dict__)                  # ... then the reference's __init__

        class DeviceTensor:          # stands in for a tensor on the GPU (there is none here): the stub only looks at .is_cuda
            is_cuda = True
        SR = 44100
        eq, comp, rev, gain = pkg.ParametricEQ(SR), pkg.Compressor(SR), pkg.NoiseShapedReverb(SR), pkg.Gain(SR)
        for proc, name in ((eq, "parametric_eq"), (comp, "compressor"), (rev, "noise_shaped_reverberation"), (gain, "gain")):
            assert proc.process_fn(DeviceTensor(), SR) == ("amd", name)
        assert pkg.signal.sosfilt_via_fsm(None, DeviceTensor()) == ("amd", "sosfilt_via_fsm")
        assert pkg.signal.lfilter_via_fsm(DeviceTensor(), None) == ("amd", "lfilter_via_fsm")
        assert pkg.signal.biquad(DeviceTensor(), None, None, SR) == ("amd", "biquad")
        assert pkg.functional.stereo_bus(DeviceTensor(), SR, None) == ("amd", "stereo_bus")
        assert seen == ["parametric_eq", "compressor", "noise_shaped_reverberation", "gain", "sosfilt_via_fsm", "lfilter_via_fsm", "biquad",
                        "stereo_bus"]
        # CPU tensors keep the reference's own path, through the same Processor objects (process_normalized passes keywords)
        x = torch.rand(2, 1, 256) * 2 - 1
        y = gain.process_normalized(x, torch.tensor([[0.5], [1.0]]))
        assert torch.allclose(y[0], x[0]) and torch.allclose(y[1], x[1] * 10 ** (24 / 20), rtol=1e-5)
        y = eq.process_normalized(x, torch.rand(2, 18))
        assert y.shape == x.shape and torch.isfinite(y).all() and len(seen) == 8
    finally:
        for k in [k for k in sys.modules if k == "dasp_pytorch" or k.startswith("dasp_pytorch.")]:
            del sys.modules[k]
        sys.modules.update(saved)
