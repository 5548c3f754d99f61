"""GPU parity of stereo_widener / stereo_panner / stereo_bus through the C ABI: reference-generated golden (fp64 run), the numpy
oracle at other shapes (ragged N, many tracks), reference shape / error conventions.
Tolerance: 1e-6 L-inf/peak for values and grad_x (pure elementwise fp32), 2e-5 of the largest entry for the reduced control gradients."""
import numpy as np
import pytest
import torch

from oracle import dasp_oracle as orc
from tests.util import linf_peak, load_golden

pytestmark = pytest.mark.gpu
SR = 44100


@pytest.fixture(scope="module")
def D():
    assert torch.cuda.is_available()
    import dasp_pytorch_amd as D
    return D


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def run(fn, x, c, w):
    xt = dev(x).requires_grad_(True)
    ct = dev(c).requires_grad_(True)
    y = fn(xt, SR, ct)
    (y * dev(w)).sum().backward()
    torch.cuda.synchronize()
    return y.detach().cpu().numpy(), xt.grad.cpu().numpy(), ct.grad.cpu().numpy()


def test_stereo_golden(D):
    g = load_golden("stereo_b2t3_n1501")
    for key, fn, x, c, w in (("wid", D.stereo_widener, "xw", "width", "ww"), ("pan", D.stereo_panner, "xp", "pan", "wp"),
                             ("bus", D.stereo_bus, "xb", "send", "wb")):
        y, gx, gc = run(fn, g[x], g[c], g[w])
        assert y.shape == g[key + "_y64"].shape and gx.shape == g[key + "_gx64"].shape and gc.shape == g[key + "_gc64"].shape
        assert linf_peak(y, g[key + "_y64"]).max() < 1e-6
        assert linf_peak(gx, g[key + "_gx64"]).max() < 1e-6
        assert np.abs(gc - g[key + "_gc64"]).max() < 2e-5 * np.abs(g[key + "_gc64"]).max()
        assert linf_peak(y, g[key + "_y32"]).max() < 1e-4            # literal north_star bar vs the reference's fp32 run


@pytest.mark.parametrize("B,T,N", [(1, 1, 1), (2, 3, 7), (1, 5, 4096), (3, 2, 4099), (2, 64, 8192), (4, 8, 131072)])
def test_stereo_shapes_vs_oracle(D, B, T, N):
    rng = np.random.default_rng(7 * B + T + N)
    xw = (rng.random((B, 2, N)) * 2 - 1).astype(np.float32); width = rng.random((B, 1)).astype(np.float32)
    xp = (rng.random((B, T, N)) * 2 - 1).astype(np.float32); pan = (rng.random((B, T)) * 0.9 + 0.05).astype(np.float32)
    xb = (rng.random((B, 2, T, N)) * 2 - 1).astype(np.float32); send = (rng.random((B, T, 1)) * 36 - 24).astype(np.float32)
    for fn, f, fv, x, c, oshape in ((D.stereo_widener, orc.stereo_widener, orc.stereo_widener_vjp, xw, width, (B, 2, N)),
                                    (D.stereo_panner, orc.stereo_panner, orc.stereo_panner_vjp, xp, pan, (B, 2, T, N)),
                                    (D.stereo_bus, orc.stereo_bus, orc.stereo_bus_vjp, xb, send, (B, 2, N))):
        w = rng.standard_normal(oshape).astype(np.float32)
        y, gx, gc = run(fn, x, c, w)
        yo = f(x, SR, c); gxo, gco = fv(x, SR, c, w)
        assert np.abs(y - yo).max() <= 2e-6 * max(np.abs(yo).max(), 1e-30)
        assert np.abs(gx - gxo).max() <= 2e-6 * max(np.abs(gxo).max(), 1e-30)
        assert np.abs(gc - gco).max() <= 5e-5 * max(np.abs(gco).max(), 1e-30)


def test_stereo_conventions(D):
    x = torch.rand(2, 2, 100, device="cuda:0")
    with pytest.raises(AssertionError):
        D.stereo_widener(torch.rand(2, 1, 100, device="cuda:0"), SR, torch.rand(2, 1, device="cuda:0"))
    with pytest.raises(AssertionError):
        D.stereo_bus(torch.rand(2, 3, 4, 100, device="cuda:0"), SR, torch.rand(2, 4, 1, device="cuda:0"))
    with pytest.raises(RuntimeError):
        D.stereo_widener(x, SR, torch.rand(3, 1, device="cuda:0"))
    x0 = x.clone()
    y = D.stereo_widener(x, SR, torch.full((2, 1), 0.5, device="cuda:0"))
    assert torch.equal(x, x0)                                     # input not mutated
    assert torch.allclose(y, x, atol=1e-6)                        # width 0.5 is the identity
    pan_half = D.stereo_panner(torch.ones(1, 1, 8, device="cuda:0"), SR, torch.full((1, 1), 0.5, device="cuda:0"))
    assert pan_half.shape == (1, 2, 1, 8) and torch.allclose(pan_half[:, 0], pan_half[:, 1], atol=1e-6)
    with pytest.raises(NotImplementedError):
        D.graphic_eq(x, SR)
    with pytest.raises(NotImplementedError):
        D.advanced_distortion(x, SR, None, None, None, None)
    # the stereo utilities compute in fp32 only: float64 input is refused rather than rounded silently (tests/test_gpu_fp64.py)
    from dasp_pytorch_amd._lib import DaspHipError
    with pytest.raises(DaspHipError, match="float64"):
        D.stereo_bus(torch.rand(1, 2, 3, 64, device="cuda:0", dtype=torch.float64), SR, torch.zeros(1, 3, 1, device="cuda:0", dtype=torch.float64))
