"""The float64 path (dasp_pytorch_amd/ops64.py, csrc/ref64.hip): the reference follows the dtype of its input (`.type_as(x)`,
dasp_pytorch/signal.py:113,119, functional.py:211), so float64 tensors mean float64 arithmetic.

  * against the reference's own fp64 outputs (tests/golden, stored as fp32: 6e-8 relative is the floor of the comparison) - far inside
    what fp32 arithmetic could reach for the parameter gradients;
  * torch.autograd.gradcheck (SURVEY section 4's fp64 oracle): every hand-derived adjoint against numerical Jacobians;
  * ops without a double-precision path refuse float64 instead of rounding it silently."""
import numpy as np
import pytest
import torch

from dasp_pytorch_amd import config

from tests.util import linf_peak, load_golden

pytestmark = pytest.mark.gpu
SR = 44100


@pytest.fixture(scope="module")
def D():
    assert torch.cuda.is_available()
    import dasp_pytorch_amd as D
    return D


def dev64(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0", torch.float64)


def test_parametric_eq_fp64_golden(D):
    g = load_golden("eq_b3c2_n12000")
    x = dev64(g["x"]).requires_grad_(True)
    cols = [dev64(g["params"][:, i]).requires_grad_(True) for i in range(18)]
    y = D.parametric_eq(x, SR, *cols)
    assert y.dtype == torch.float64
    (y * dev64(g["w"])).sum().backward()
    gp = torch.stack([c.grad for c in cols], 1).cpu().numpy()
    # the reference's FSM is circular: for these filters (impulse responses decayed within the signal) it equals the recursion to ~1e-9
    assert linf_peak(y.detach().cpu().numpy(), g["y64"]).max() < 5e-7
    assert linf_peak(x.grad.cpu().numpy(), g["gx64"]).max() < 5e-7
    assert linf_peak(gp, g["gp64"]).max() < 2e-6            # fp32 kernels: 1e-5 .. 1e-4 here


def test_sosfilt_and_lfilter_fp64_golden(D):
    g = load_golden("sos_b2c2_n6000_s3")
    x = dev64(g["x"]).requires_grad_(True)
    sos = dev64(g["sos"]).requires_grad_(True)
    y = D.signal.sosfilt_via_fsm(sos, x)
    (y * dev64(g["w"])).sum().backward()
    assert y.dtype == torch.float64 and sos.grad.dtype == torch.float64
    assert linf_peak(y.detach().cpu().numpy(), g["y64"]).max() < 5e-7
    assert linf_peak(x.grad.cpu().numpy(), g["gx64"]).max() < 5e-7
    assert linf_peak(sos.grad.cpu().numpy(), g["gsos64"]).max() < 2e-6
    g = load_golden("lfilter_b3_n9000")
    for key in ("onepole", "iir2", "fir"):
        x = dev64(g["x"]).requires_grad_(True)
        b = dev64(g["b_" + key]).requires_grad_(True)
        a = dev64(g["a_" + key]).requires_grad_(True) if key != "fir" else None
        y = D.signal.lfilter_via_fsm(x, b, a)
        (y * dev64(g["w"])).sum().backward()
        assert linf_peak(y.detach().cpu().numpy(), g[key + "_y64"]).max() < 5e-7, key
        assert linf_peak(b.grad.cpu().numpy(), g[key + "_gb64"]).max() < 2e-6, key
        if a is not None:
            assert linf_peak(a.grad.cpu().numpy(), g[key + "_ga64"]).max() < 2e-6, key


@pytest.mark.parametrize("name", ["comp_b3c2_n12000", "comp_b2c1_n20011_look7"])
def test_compressor_fp64_golden(D, name):
    g = load_golden(name)
    x = dev64(g["x"]).requires_grad_(True)
    cols = [dev64(g["params"][:, i]).requires_grad_(True) for i in range(6)]
    y = D.compressor(x, SR, *cols, lookahead_samples=int(g["lookahead"]))
    (y * dev64(g["w"])).sum().backward()
    assert y.dtype == torch.float64
    assert linf_peak(y.detach().cpu().numpy(), g["y64"]).max() < 5e-7
    assert linf_peak(x.grad.cpu().numpy(), g["gx64"]).max() < 5e-7
    gp = torch.stack([c.grad for c in cols], 1).cpu().numpy()
    for j in range(6):
        ref = g["gp64"][:, j]
        assert np.abs(gp[:, j] - ref).max() <= 2e-6 * max(np.abs(ref).max(), 1e-12), j


def test_gain_distortion_fp64_golden(D):
    g = load_golden("gain_dist_cfg1")
    for fn, key, pre in ((D.gain, "gain_db", "gain"), (D.distortion, "drive_db", "dist")):
        x = dev64(g["x"]).requires_grad_(True)
        c = dev64(g[key]).requires_grad_(True)
        y = fn(x, SR, c)
        (y * dev64(g["w"])).sum().backward()
        assert y.dtype == torch.float64
        assert linf_peak(y.detach().cpu().numpy(), g[pre + "_y64"]).max() < 5e-7
        assert linf_peak(x.grad.cpu().numpy(), g[pre + "_gx64"]).max() < 5e-7
        assert np.abs(c.grad.cpu().numpy() - g[pre + "_gp64"]).max() < 5e-7 * np.abs(g[pre + "_gp64"]).max()


def test_gradcheck_all_fp64_ops(D):
    """torch.autograd.gradcheck on small float64 inputs: analytical (hand-derived adjoint kernels) against numerical Jacobians."""
    gen = torch.Generator(device="cuda:0").manual_seed(0)
    rnd = lambda *s: torch.rand(*s, device="cuda:0", dtype=torch.float64, generator=gen)
    B, C, N = 2, 2, 48
    x = (rnd(B, C, N) * 2 - 1).requires_grad_(True)
    R = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
         (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
    eqc = [(rnd(B) * (hi - lo) + lo).requires_grad_(True) for lo, hi in R]
    assert torch.autograd.gradcheck(lambda x, *c: D.parametric_eq(x, SR, *c), (x, *eqc), eps=1e-6, atol=1e-7, rtol=1e-5, nondet_tol=1e-12)
    sos = torch.tensor([[[0.3, -0.2, 0.1, 1.2, -0.9, 0.4], [1.0, 0.5, 0.25, 0.8, 0.3, 0.2], [0.7, 0.0, -0.3, 1.0, -1.2, 0.5]]],
                       device="cuda:0", dtype=torch.float64).repeat(B, 1, 1)
    sos = (sos * (1 + 0.1 * rnd(B, 3, 6))).requires_grad_(True)
    assert torch.autograd.gradcheck(D.signal.sosfilt_via_fsm, (sos, x), eps=1e-6, atol=1e-7, rtol=1e-5, nondet_tol=1e-12)
    # compressor / expander: away from the knee edges and the eps clamp, where the map is smooth
    xs = ((rnd(B, C, N) * 0.5 + 0.1) * torch.where(rnd(B, C, N) > 0.5, 1.0, -1.0)).requires_grad_(True)
    dyn = [torch.tensor(v, device="cuda:0", dtype=torch.float64).requires_grad_(True) for v in
           ([-30.0, -18.0], [4.0, 2.5], [5.0, 12.0], [50.0, 20.0], [6.0, 3.0], [2.0, 0.5])]
    for fn, look in ((D.compressor, 0), (D.compressor, 3), (D.expander, 0)):
        assert torch.autograd.gradcheck(lambda x, *c: fn(x, SR, *c, lookahead_samples=look), (xs, *dyn), eps=1e-6, atol=1e-7, rtol=1e-4,
                                        nondet_tol=1e-12), (fn.__name__, look)
    gdb = (rnd(B) * 48 - 24).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda x, g: D.gain(x, SR, g), (x, gdb), eps=1e-6, atol=1e-7, rtol=1e-5, nondet_tol=1e-12)
    ddb = (rnd(B * C) * 24).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda x, g: D.distortion(x, SR, g), (x, ddb), eps=1e-6, atol=1e-7, rtol=1e-5, nondet_tol=1e-12)
    b = (rnd(B, 2) - 0.5).requires_grad_(True)
    a = torch.cat([torch.ones(B, 1, device="cuda:0", dtype=torch.float64), -0.9 * rnd(B, 1)], 1).requires_grad_(True)
    x1 = (rnd(B, 1, N) * 2 - 1).requires_grad_(True)
    assert torch.autograd.gradcheck(D.signal.lfilter_via_fsm, (x1, b, a), eps=1e-6, atol=1e-7, rtol=1e-5, nondet_tol=1e-12)
    # K = 5 and K = 11 coefficients (csrc/lfilter.hip; a0 != 1: the normalisation is differentiated by torch), several chunks of time
    import os
    config.plan.lfilter_chunk = 13
    try:
        for K in (5, 11):
            bl = ((rnd(B, K) - 0.5) * 0.5).requires_grad_(True)
            al = torch.cat([1.0 + rnd(B, 1), (rnd(B, K - 1) - 0.5) * (0.8 / K)], 1).requires_grad_(True)
            assert torch.autograd.gradcheck(D.signal.lfilter_via_fsm, (x1, bl, al), eps=1e-6, atol=1e-7, rtol=1e-5, nondet_tol=1e-10), K
        assert torch.autograd.gradcheck(lambda x_, b_: D.signal.lfilter_via_fsm(x_, b_, None), (x1, ((rnd(1, 7) - 0.5)).requires_grad_(True)), eps=1e-6,
                                        atol=1e-7, rtol=1e-5, nondet_tol=1e-10)                      # FIR shared by the batch
    finally:
        config.plan.lfilter_chunk = 0


def test_ops_without_a_double_path_refuse_float64(D, monkeypatch):
    from dasp_pytorch_amd._lib import DaspHipError
    x = torch.rand(2, 2, 4096, device="cuda:0", dtype=torch.float64)
    one = lambda v: torch.full((2,), v, device="cuda:0")
    with pytest.raises(DaspHipError, match="float64"):
        D.noise_shaped_reverberation(x, SR, *[one(0.5)] * 25, num_samples=1024, num_bandpass_taps=63)
    with pytest.raises(DaspHipError, match="float64"):
        D.stereo_widener(x, SR, one(0.3).reshape(2, 1))
    with pytest.raises(DaspHipError, match="float64"):
        D.losses.MultiResolutionSTFTLoss()(x, x)
    monkeypatch.setattr(config.plan, "fp64_as_fp32", True)            # the explicit opt-in: cast, compute in fp32, cast back
    y = D.stereo_widener(x, SR, one(0.3).reshape(2, 1))
    assert y.dtype == torch.float64
