"""GPU parity of the multi-resolution STFT loss (csrc/stftloss.hip) against the numpy oracle (oracle/dasp_oracle.py:mrstft_loss,
the restated auraloss algorithm, itself checked against a torch.stft implementation in tests/test_oracle_cpu.py).
Tolerance: 2e-5 relative on the loss (fp32 FFTs + fp32 partial sums of ~1e6 terms; measured ~1e-8). The gradient of a
log-magnitude L1 loss is ill-conditioned wherever a predicted magnitude is small (weight 1/|P|, direction P/|P|, sign of a
difference): a torch.stft implementation of the same loss in fp32 is 1e-4 ... 3e-3 away from its fp64 run on these inputs, the
kernels 1.5e-4 ... 6e-3. Bounds on such inputs: 1e-2 in relative L2 norm, 2e-2 of the largest entry; where the gradient is
well-conditioned (test_mrstft_gradient_on_well_conditioned_input) the bound is 1e-4."""
import numpy as np
import pytest
import torch

from oracle import dasp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D():
    assert torch.cuda.is_available()
    import dasp_pytorch_amd as D
    return D


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("B,C,N,res", [(2, 2, 6000, None), (1, 1, 1500, None), (3, 1, 20000, None), (1, 2, 4097, ((256, 64, 256), (64, 16, 48))),
                                       (2, 1, 9000, ((4096, 1024, 4096), (8, 2, 8), (512, 128, 500)))])
def test_mrstft_vs_oracle(D, B, C, N, res):
    rng = np.random.default_rng(N)
    a = (rng.standard_normal((B, C, N)) * 0.3).astype(np.float32)
    b = (0.6 * a + 0.2 * rng.standard_normal((B, C, N))).astype(np.float32)
    kw = {} if res is None else dict(fft_sizes=[r[0] for r in res], hop_sizes=[r[1] for r in res], win_lengths=[r[2] for r in res])
    at = dev(a).requires_grad_(True)
    loss = D.losses.MultiResolutionSTFTLoss(**kw)(at, dev(b))
    (3.0 * loss).backward()
    lo = orc.mrstft_loss(a, b, **({} if res is None else {"resolutions": res}))
    go = 3.0 * orc.mrstft_loss_vjp(a, b, **({} if res is None else {"resolutions": res}))
    assert abs(float(loss.detach()) - lo) < 2e-5 * abs(lo)
    g = at.grad.cpu().numpy()
    assert np.linalg.norm(g - go) < 1e-2 * np.linalg.norm(go)
    assert np.abs(g - go).max() < 2e-2 * np.abs(go).max()


@pytest.mark.parametrize("N,res,floor,noise,tol", [(3000, ((256, 64, 256), (64, 16, 64)), 1e-3, 1e-3, 1e-4), (6000, ((512, 128, 400), (128, 32, 128)), 1e-3, 1e-3, 1e-4),
                                                   (9000, ((2048, 512, 1200), (1024, 256, 600)), 3e-4, 1e-5, 3e-4),
                                                   (7000, ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)), 3e-4, 1e-5, 3e-4)])
def test_mrstft_gradient_on_well_conditioned_input(D, N, res, floor, noise, tol):
    """The gradient where it is well-conditioned: the prediction is 1.5 x the target plus a small perturbation, so the sign of every
    log-magnitude difference is fixed (log 1.5 > 0; the draw is checked with the oracle's own spectra to keep every difference above
    0.1 and every predicted magnitude above `floor` of the largest one; the last two cases run the 512 / 1024 / 2048-point frames of the default
    resolutions, i.e. the kernels with 1, 2 and 4 waves per frame - thousands of bins per frame, so the smallest one is 3e-4 of the largest
    and the bound is 3e-4). There the kernels are held to 1e-4 in relative L2 norm and of the
    largest entry (measured 4e-6 .. 2e-5) - a wrong window, padding or scaling term would be off by orders of magnitude more."""
    for seed in range(40):
        rng = np.random.default_rng(1000 * N + seed)
        b = (rng.standard_normal((1, 1, N)) * 0.3).astype(np.float32)
        a = (1.5 * b + noise * rng.standard_normal((1, 1, N))).astype(np.float32)
        ok = True
        for n_fft, hop, win in res:
            pm = orc._stft_mag(a[0], n_fft, hop, win, 1e-8, np.float64)[0]
            tm = orc._stft_mag(b[0], n_fft, hop, win, 1e-8, np.float64)[0]
            ok = ok and pm.min() > floor * pm.max() and (np.log(pm) - np.log(tm)).min() > 0.1
        if ok:
            break
    else:
        pytest.skip("no well-conditioned draw found")
    kw = dict(fft_sizes=[r[0] for r in res], hop_sizes=[r[1] for r in res], win_lengths=[r[2] for r in res])
    at = dev(a).requires_grad_(True)
    loss = D.losses.MultiResolutionSTFTLoss(**kw)(at, dev(b))
    loss.backward()
    lo = orc.mrstft_loss(a, b, resolutions=res)
    go = orc.mrstft_loss_vjp(a, b, resolutions=res)
    g = at.grad.cpu().numpy()
    assert abs(float(loss.detach()) - lo) < 2e-5 * abs(lo)
    e2, einf = np.linalg.norm(g - go) / np.linalg.norm(go), np.abs(g - go).max() / np.abs(go).max()
    print("mrstft well-conditioned gradient error: rel L2 %.2e, max %.2e" % (e2, einf))
    assert e2 < tol and einf < tol


def test_mrstft_conventions(D):
    x = torch.rand(2, 1, 3000, device="cuda:0")
    fn = D.losses.MultiResolutionSTFTLoss()
    assert float(fn(x, x)) < 1e-5                                     # identical signals (the two spectra come out of one packed transform: ~1e-7, not 0)
    xg = x.clone().requires_grad_(True)
    fn(xg, x).backward()
    assert torch.isfinite(xg.grad).all()
    y = torch.rand(2, 1, 3000, device="cuda:0")
    l1, l2 = fn(x, y), D.losses.mrstft_loss(x, y)
    assert float(l1) == float(l2) and l1.dtype == x.dtype and l1.ndim == 0
    from dasp_pytorch_amd._lib import DaspHipError
    with pytest.raises(DaspHipError):
        fn(torch.rand(1, 1, 600, device="cuda:0"), torch.rand(1, 1, 600, device="cuda:0"))        # 2048-point frames need more than 1024 samples
    with pytest.raises(RuntimeError):
        fn(x, y[:, :, :100])
    with pytest.raises(DaspHipError):
        fn(x.cpu(), y.cpu())


def test_single_resolution_stft_loss(D):
    """losses.STFTLoss (auraloss.freq.STFTLoss with its defaults: fft 1024, hop 256, window 1024; the reference's
    examples/blind_estimation.py:141) = one resolution of the same kernels: value and gradient against the oracle."""
    rng = np.random.default_rng(12)
    B, C, N = 3, 2, 20000
    x = (rng.random((B, C, N)) * 0.8 - 0.4).astype(np.float32)
    y = (x + 0.05 * rng.standard_normal((B, C, N))).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    loss = D.losses.STFTLoss()(xt, torch.from_numpy(y).cuda())
    loss.backward()
    res = ((1024, 256, 1024),)
    lo = orc.mrstft_loss(x, y, res)
    go = orc.mrstft_loss_vjp(x, y, res)
    assert abs(float(loss) - lo) < 2e-5 * abs(lo)
    g = xt.grad.cpu().numpy()                   # (sign(log P - log T) flips where the two magnitudes meet: a norm, not the largest entry)
    assert np.linalg.norm(g - go) < 1e-2 * np.linalg.norm(go)


def _torch_mrstft(p, t, resolutions, eps=1e-8):
    """auraloss.freq.MultiResolutionSTFTLoss with its defaults, restated on torch.stft (float64, CPU, autograd for both arguments)."""
    total = 0.0
    for n_fft, hop, win in resolutions:
        w = torch.hann_window(win, dtype=p.dtype)
        mag = lambda v: torch.sqrt(torch.clamp(torch.view_as_real(torch.stft(v.reshape(-1, v.shape[-1]), n_fft, hop, win, w, return_complex=True)).pow(2).sum(-1), min=eps))
        P, T = mag(p), mag(t)
        total = total + torch.norm(T - P, p="fro") / torch.norm(T, p="fro") + (torch.log(P) - torch.log(T)).abs().mean()
    return total / len(resolutions)


@pytest.mark.parametrize("res", [((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)), ((256, 64, 256),)])
def test_gradient_for_both_arguments(D, res):
    """auraloss differentiates input and target; so does this loss (dasp_mrstft_backward_target: the backward kernels with the two signals
    swapped plus the norm term of the spectral convergence). Both gradients against torch.stft + autograd in float64 on a draw whose
    log-magnitude differences keep their sign (prediction = 1.5 x target + a little noise), and the plain case target.requires_grad =
    False still returns None for it."""
    rng = np.random.default_rng(3)
    N = 9000
    t = (rng.standard_normal((2, 1, N)) * 0.3).astype(np.float32)
    p = (1.5 * t + 0.003 * rng.standard_normal((2, 1, N))).astype(np.float32)
    kw = dict(fft_sizes=[r[0] for r in res], hop_sizes=[r[1] for r in res], win_lengths=[r[2] for r in res])
    pt, tt = dev(p).requires_grad_(True), dev(t).requires_grad_(True)
    loss = D.losses.MultiResolutionSTFTLoss(**kw)(pt, tt)
    loss.backward()
    pc, tc = torch.from_numpy(p).double().requires_grad_(True), torch.from_numpy(t).double().requires_grad_(True)
    ref = _torch_mrstft(pc, tc, res)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-5 * abs(float(ref.detach()))
    for name, g, go in (("input", pt.grad, pc.grad), ("target", tt.grad, tc.grad)):
        g, go = g.cpu().double(), go
        e2 = float((g - go).norm() / go.norm())
        print(f"mrstft gradient w.r.t. {name}: rel L2 {e2:.2e}")
        assert e2 < 2e-3, (name, e2)
    t2 = dev(t)
    p2 = dev(p).requires_grad_(True)
    D.losses.MultiResolutionSTFTLoss(**kw)(p2, t2).backward()
    assert t2.grad is None and float((p2.grad - pt.grad).abs().max()) <= 1e-6 * float(pt.grad.abs().max())
