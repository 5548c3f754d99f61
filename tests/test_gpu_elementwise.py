"""GPU parity of gain / distortion (BASELINE config 1 and larger) through the C ABI.
Tolerance: 1e-6 L-inf/peak vs the fp64 reference for values and grad_x (pure elementwise fp32),
1e-5 of the largest entry for the reduced control gradients (signed fp32 sums)."""
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


def test_config1_golden(D):
    g = load_golden("gain_dist_cfg1")
    for name, fn, ctl in (("gain", D.gain, "gain_db"), ("dist", D.distortion, "drive_db")):
        x = dev(g["x"]).requires_grad_(True)
        c = dev(g[ctl]).requires_grad_(True)
        y = fn(x, SR, c)
        (y * dev(g["w"])).sum().backward()
        assert linf_peak(y.detach().cpu().numpy(), g[name + "_y64"]).max() < 1e-6
        assert linf_peak(x.grad.cpu().numpy(), g[name + "_gx64"]).max() < 1e-6
        # the control gradient is a signed 16384-term fp32 sum: tolerance relative to the largest entry
        assert np.abs(c.grad.cpu().numpy() - g[name + "_gp64"]).max() < 1e-5 * np.abs(g[name + "_gp64"]).max()
        # literal north_star bar vs the reference's own fp32 output
        assert linf_peak(y.detach().cpu().numpy(), g[name + "_y32"]).max() < 1e-4


def test_per_sample_drive_golden_and_shapes(D):
    """distortion with one drive value per sample: the reference's drive_db.view(bs, chs, -1) with bs*chs*seq_len values (functional.py:78).
    Reference-generated golden, then odd sizes (scalar tail path) against the oracle; counts the reference's view / broadcast rejects raise."""
    g = load_golden("dist_sample_b2c2_n3001")
    x = dev(g["x"]).requires_grad_(True)
    d = dev(g["drive_db"]).requires_grad_(True)
    y = D.distortion(x, SR, d)
    (y * dev(g["w"])).sum().backward()
    assert linf_peak(y.detach().cpu().numpy(), g["y64"]).max() < 1e-6 and linf_peak(y.detach().cpu().numpy(), g["y32"]).max() < 1e-4
    assert linf_peak(x.grad.cpu().numpy(), g["gx64"]).max() < 2e-6
    assert d.grad.shape == d.shape and linf_peak(d.grad.cpu().numpy(), g["gp64"]).max() < 2e-6
    rng = np.random.default_rng(3)
    for B, C, N in ((1, 1, 2), (2, 3, 5), (1, 2, 4096), (3, 1, 10001)):
        xs = (rng.random((B, C, N)) * 2 - 1).astype(np.float32); ds = (rng.random((B, C, N)) * 24).astype(np.float32)
        ws = rng.standard_normal((B, C, N)).astype(np.float32)
        xt = dev(xs).requires_grad_(True); dt = dev(ds.reshape(-1)).requires_grad_(True)          # any shape with the right count, as view() takes
        (D.distortion(xt, SR, dt) * dev(ws)).sum().backward()
        gxo, gdo = orc.distortion_vjp(xs, SR, ds, ws)
        assert np.abs(xt.grad.cpu().numpy() - gxo).max() < 2e-6 * max(1.0, np.abs(gxo).max())
        assert dt.grad.shape == dt.shape and np.abs(dt.grad.cpu().numpy().reshape(B, C, N) - gdo).max() < 2e-6 * max(1.0, np.abs(gdo).max())
    with pytest.raises(RuntimeError):
        D.distortion(torch.zeros(2, 2, 8, device="cuda:0"), SR, torch.zeros(5, device="cuda:0"))       # view(2, 2, -1) fails
    with pytest.raises(RuntimeError):
        D.distortion(torch.zeros(2, 2, 8, device="cuda:0"), SR, torch.zeros(8, device="cuda:0"))       # views to (2, 2, 2): no broadcast against 8


@pytest.mark.parametrize("B,C,N", [(1, 1, 1), (2, 3, 5), (3, 2, 8191), (2, 2, 8192), (1, 2, 8193), (4, 2, 100000), (70000, 1, 64), (256, 2, 131072)])
def test_shapes_vs_oracle(D, B, C, N):
    rng = np.random.default_rng(N + B)
    x = (rng.random((B, C, N)) * 2 - 1).astype(np.float32)
    w = rng.standard_normal((B, C, N)).astype(np.float32)
    gain_db = (rng.random(B) * 48 - 24).astype(np.float32)
    drive_db = (rng.random(B * C) * 24).astype(np.float32)
    check_all = B * C * N < 5_000_000      # full-size case: spot-check a few rows against the oracle
    sel = slice(None) if check_all else slice(0, 3)
    for fn, ctl, f, fv in ((D.gain, gain_db, orc.gain, orc.gain_vjp), (D.distortion, drive_db, orc.distortion, orc.distortion_vjp)):
        xt = dev(x).requires_grad_(True); ct = dev(ctl).requires_grad_(True)
        y = fn(xt, SR, ct)
        (y * dev(w)).sum().backward()
        cs = ctl[sel] if fn is D.gain else ctl.reshape(B, C)[sel].reshape(-1)
        yo = f(x[sel], SR, cs)
        gxo, gco = fv(x[sel], SR, cs, w[sel])
        assert np.abs(y.detach().cpu().numpy()[sel] - yo).max() < 2e-6 * max(1.0, np.abs(yo).max())
        assert np.abs(xt.grad.cpu().numpy()[sel] - gxo).max() < 2e-6 * max(1.0, np.abs(gxo).max())
        gc = ct.grad.cpu().numpy() if fn is D.gain else ct.grad.cpu().numpy().reshape(B, C)
        ec = np.abs(gc[sel].reshape(-1) - np.asarray(gco).reshape(-1)).max() / np.abs(gco).max()
        from tests.util import record
        record(f"elementwise_shapes[{fn.__name__},{B},{C},{N}]", gctl=ec)
        assert ec < 1e-4, ec
        assert torch.isfinite(y).all() and torch.isfinite(xt.grad).all() and torch.isfinite(ct.grad).all()


def test_semantics(D):
    x = torch.rand(2, 2, 1000, device="cuda:0") * 2 - 1
    x0 = x.clone()
    # 0 dB gain = identity, inputs never mutated, fp64 follows x, integer controls are legal
    assert torch.equal(D.gain(x, SR, torch.zeros(2, device="cuda:0")), x) and torch.equal(x, x0)
    assert D.gain(x.double(), SR, torch.zeros(2, device="cuda:0")).dtype == torch.float64
    assert torch.allclose(D.gain(x, SR, torch.tensor([6, -6], device="cuda:0")), x * (10 ** (torch.tensor([6., -6.], device="cuda:0") / 20)).view(2, 1, 1), rtol=1e-6)
    # gain takes (bs,), distortion needs bs*chs values (reference view semantics): wrong sizes raise RuntimeError
    with pytest.raises(RuntimeError):
        D.gain(x, SR, torch.zeros(3, device="cuda:0"))
    with pytest.raises(RuntimeError):
        D.distortion(x, SR, torch.zeros(2, device="cuda:0"))
    y = D.distortion(x, SR, torch.zeros(2, 2, device="cuda:0"))
    assert torch.allclose(y, torch.tanh(x), atol=2e-7)
