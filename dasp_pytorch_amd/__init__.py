"""dasp_pytorch_amd -- MI355X-native hot path of dasp_pytorch.functional."""
from .functional import parametric_eq  # noqa: F401
from . import signal  # noqa: F401
