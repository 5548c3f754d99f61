"""Independent fp64 check: the same LTI systems evaluated as true recursions with SciPy.

TEST INFRASTRUCTURE ONLY. The reference filters with a frequency-sampling FFT method
(dasp_pytorch/signal.py:136-166) that equals the recursion to <= 1e-13 once the impulse response
has decayed within the signal; for short signals, where the
reference's circular convolution time-aliases, these recursions are the ground truth the kernels
(which are recursions) are held to."""
import numpy as np
from scipy import signal as _ss


def sosfilt_ref(sos, x):
    """sos (bs or 1, S, 6) rows [b0 b1 b2 a0 a1 a2] (any a0), x (bs, ..., T) -> y fp64."""
    sos = np.asarray(sos, np.float64)
    x = np.asarray(x, np.float64)
    y = np.empty_like(x)
    for b in range(x.shape[0]):
        s = sos[b if sos.shape[0] > 1 else 0].copy()
        s = s / s[:, 3:4]
        y[b] = _ss.sosfilt(s, x[b], axis=-1)
    return y


def sosfilt_vjp_ref(sos, gy):
    """grad wrt x of sum(y * gy): the adjoint of a causal LTI map is the same filter run backwards in time."""
    gy = np.asarray(gy, np.float64)
    return sosfilt_ref(sos, gy[..., ::-1])[..., ::-1]


def one_pole_ref(x, alpha):
    """g[n] = (1 - alpha) x[n] + alpha g[n-1] per batch row (the compressor's smoothing filter,
    dasp_pytorch/functional.py:372-380). x (bs, T), alpha (bs,)."""
    x = np.asarray(x, np.float64)
    y = np.empty_like(x)
    for b in range(x.shape[0]):
        a = float(alpha[b])
        y[b] = _ss.lfilter([1.0 - a], [1.0, -a], x[b])
    return y
