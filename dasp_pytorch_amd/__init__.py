"""dasp_pytorch_amd -- MI355X-native hot path of dasp_pytorch.functional."""
from . import functional, signal  # noqa: F401
from .functional import distortion, gain, parametric_eq  # noqa: F401
