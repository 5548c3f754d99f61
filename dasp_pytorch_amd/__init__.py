"""dasp_pytorch_amd -- MI355X-native hot path of dasp_pytorch.functional."""
from . import chain, config, functional, losses, modules, signal  # noqa: F401
from .functional import (advanced_distortion, compressor, distortion, expander, gain, graphic_eq, noise_shaped_reverberation,  # noqa: F401
                         parametric_eq, stereo_bus, stereo_panner, stereo_widener)
from .modules import Compressor, Distortion, Expander, Gain, NoiseShapedReverb, ParametricEQ, Processor  # noqa: F401
