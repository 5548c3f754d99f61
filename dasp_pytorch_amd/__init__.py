"""dasp_pytorch_amd -- MI355X-native hot path of dasp_pytorch.functional."""
from . import functional, modules, signal  # noqa: F401
from .functional import compressor, distortion, expander, gain, noise_shaped_reverberation, parametric_eq  # noqa: F401
from .modules import Compressor, Distortion, Expander, Gain, NoiseShapedReverb, ParametricEQ  # noqa: F401
