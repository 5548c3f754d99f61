"""GPU parity of noise_shaped_reverberation against reference-generated goldens (same noise) and the numpy oracle.
Tolerance: 2e-5 L-inf/peak for y / grad_x (fp32 FFTs of length 2^12..2^17; SURVEY measured 4.9e-6 for an fp32 FFT
restatement), 1e-4 of the largest entry for the 25 control gradients (measured: y / grad_x <= 8.2e-7, control gradients <= 1.9e-6)."""
import numpy as np
import pytest
import torch

from oracle import dasp_oracle as orc
from tests.test_oracle_cpu import _reverb_noise
from tests.util import linf_peak, load_golden, record

pytestmark = pytest.mark.gpu


class reverb_plan:
    """dasp_reverb_plan(chunk, weight_limit, band_split) for the duration of a block (-1 = the planner's own choice): the explicit C-ABI
    arguments that replaced rounds 2 - 4's DASP_REVERB_* environment switches."""

    def __init__(self, chunk=-1, weight_limit=-1.0, band_split=-1):
        self.args = (chunk, weight_limit, band_split)

    def __enter__(self):
        from dasp_pytorch_amd import _lib
        assert _lib.lib().dasp_reverb_plan(*self.args) == 0

    def __exit__(self, *exc):
        from dasp_pytorch_amd import _lib
        _lib.lib().dasp_reverb_plan(-1, -1.0, -1)
SR = 44100
CTL_TOL = 1e-4      # the 25 control gradients, of the largest entry: the north_star bar (measured 3e-9 .. 1.9e-6, profiles/r03/parity_measured.jsonl)


@pytest.fixture(scope="module")
def D():
    assert torch.cuda.is_available()
    import dasp_pytorch_amd as D
    return D


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def run(D, x, p, w, noise, L, taps):
    xt = dev(x).requires_grad_(True)
    cols = [dev(p[:, i]).requires_grad_(True) for i in range(25)]
    y = D.noise_shaped_reverberation(xt, SR, *cols, num_samples=L, num_bandpass_taps=taps, noise=None if noise is None else dev(noise))
    (y * dev(w)).sum().backward()
    torch.cuda.synchronize()
    return y.detach().cpu().numpy(), xt.grad.cpu().numpy(), torch.stack([c.grad for c in cols], 1).cpu().numpy()


@pytest.mark.parametrize("name", ["rev_b2c2_n6000_l2048_t127", "rev_b1c1_n5000_l1000_t63", "rev_b1c2_n20000_default"])
def test_reverb_golden(D, name):
    g = load_golden(name)
    noise = _reverb_noise(g)
    y, gx, gp = run(D, g["x"], g["params"], g["w"], noise, int(g["L"]), int(g["taps"]))
    assert y.shape == g["y64"].shape and gx.shape == g["gx64"].shape          # mono in -> stereo out, grad_x mono
    record(f"reverb_golden[{name}]", y=linf_peak(y, g["y64"]).max(), gx=linf_peak(gx, g["gx64"]).max(), gctl=linf_peak(gp, g["gp64"]).max())
    assert linf_peak(y, g["y64"]).max() < 2e-5
    assert linf_peak(gx, g["gx64"]).max() < 2e-5
    assert linf_peak(gp, g["gp64"]).max() < 1e-4
    assert linf_peak(y, g["y32"]).max() < 1e-4                                # literal north_star bar vs the reference's fp32 run


def test_reverb_seed_reproduces_reference_noise_stream(D):
    """Default noise path = the reference's: global CPU generator, same call (functional.py:548)."""
    g = load_golden("rev_b2c2_n6000_l2048_t127")
    torch.manual_seed(int(g["noise_seed"]))
    x = dev(g["x"]); cols = [dev(g["params"][:, i]) for i in range(25)]
    y = D.noise_shaped_reverberation(x, SR, *cols, num_samples=int(g["L"]), num_bandpass_taps=int(g["taps"]))
    e = linf_peak(y.cpu().numpy(), g["y64"]).max()
    if e > 1e-3:
        pytest.skip("this torch build's CPU generator does not reproduce the golden's noise stream")
    assert e < 2e-5


# The long convolution runs n1 = 2 nextpow2(L) point four-step transforms with column length NA = n1 / 512: the L values below walk NA
# through 8, 16, ..., 4096, i.e. every combination of radix-8 passes and radix-2/4 tail of the column transform (fft_lds.hpp col_fft),
# with odd and even block counts (zero partner in the last pair, overlap carried across pairs).
@pytest.mark.parametrize("B,C,N,L,taps", [(1, 2, 1, 64, 15), (2, 1, 100, 256, 31), (1, 2, 5000, 300, 63), (3, 2, 70000, 65536, 1023), (2, 2, 262144, 65536, 1023),
                                          (1, 2, 9000, 3000, 63), (2, 2, 21000, 5000, 255), (1, 1, 50000, 10000, 1023), (1, 2, 9000, 20000, 127),
                                          (1, 2, 300000, 100000, 63), (1, 1, 30000, 200000, 255), (1, 2, 40000, 400000, 63), (1, 2, 40000, 1000000, 63),
                                          (1, 2, 4000, 500, 3585),
                                          # more pairs of blocks: the overlap carried across pairs, an odd block count (zero partner), mono input
                                          (1, 2, 61000, 8000, 63), (2, 1, 110000, 8192, 127), (1, 2, 32768, 8192, 31), (1, 2, 120000, 30000, 255)])
def test_reverb_shapes_vs_oracle(D, B, C, N, L, taps):
    rng = np.random.default_rng(N + L)
    x = (rng.random((B, C, N)) * 2 - 1).astype(np.float32)
    w = rng.standard_normal((B, 2, N)).astype(np.float32)
    p = rng.random((B, 25)).astype(np.float32)
    noise = rng.standard_normal((2 * B, 12, L + taps - 1)).astype(np.float32)
    y, gx, gp = run(D, x, p, w, noise, L, taps)
    pd = p.astype(np.float64)
    yo = orc.noise_shaped_reverberation(x, SR, pd[:, :12], pd[:, 12:24], pd[:, 24], noise, L, taps)
    gxo, gg, gd, gm = orc.noise_shaped_reverberation_vjp(x, SR, pd[:, :12], pd[:, 12:24], pd[:, 24], noise, w, L, taps)
    assert np.abs(y - yo).max() < 3e-5 * max(np.abs(yo).max(), 1e-6)
    assert np.abs(gx - gxo).max() < 3e-5 * max(np.abs(gxo).max(), 1e-6)
    gpo = np.concatenate([gg, gd, gm[:, None]], 1)
    record(f"reverb_shapes[{B},{C},{N},{L},{taps}]", y=np.abs(y - yo).max() / max(np.abs(yo).max(), 1e-6), gx=np.abs(gx - gxo).max() / max(np.abs(gxo).max(), 1e-6),
           gctl=np.abs(gp - gpo).max() / np.abs(gpo).max())
    assert np.abs(gp - gpo).max() < CTL_TOL * np.abs(gpo).max()


def test_config4_full_size_sampled_items(D):
    """BASELINE config 4, noise_shaped_reverberation at (128, 2, 262144) with the default 65536-tap impulse responses and 1023-tap filter
    bank: the launch whose scratch sizes, XCD tile map and band-split planner depend on the batch size. Three sampled items (first, middle,
    last) of the full launch against the oracle on the same noise: y, grad x and the 25 control gradients (items are independent)."""
    B, C, N, L, taps = 128, 2, 262144, 65536, 1023
    gen = torch.Generator(device="cuda:0").manual_seed(44)
    x = torch.rand(B, C, N, device="cuda:0", generator=gen) * 2 - 1
    w = torch.randn(B, 2, N, device="cuda:0", generator=gen)
    p = torch.rand(B, 25, device="cuda:0", generator=gen)
    noise = torch.randn(2 * B, 12, L + taps - 1, device="cuda:0", generator=gen)
    xt = x.clone().requires_grad_(True)
    cols = [p[:, i].clone().requires_grad_(True) for i in range(25)]
    y = D.noise_shaped_reverberation(xt, SR, *cols, num_samples=L, num_bandpass_taps=taps, noise=noise)
    y.backward(w)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and torch.isfinite(xt.grad).all() and all(torch.isfinite(c.grad).all() for c in cols)
    gp = torch.stack([c.grad for c in cols], 1)
    idx = [0, 77, 127]
    nidx = [2 * i + k for i in idx for k in (0, 1)]
    xs, ws, ps, ns = x[idx].cpu().numpy(), w[idx].cpu().numpy(), p[idx].cpu().numpy().astype(np.float64), noise[nidx].cpu().numpy()
    yo = orc.noise_shaped_reverberation(xs, SR, ps[:, :12], ps[:, 12:24], ps[:, 24], ns, L, taps)
    gxo, gg, gd, gm = orc.noise_shaped_reverberation_vjp(xs, SR, ps[:, :12], ps[:, 12:24], ps[:, 24], ns, ws, L, taps)
    gpo = np.concatenate([gg, gd, gm[:, None]], 1)
    ey, egx = linf_peak(y.detach()[idx].cpu().numpy(), yo), linf_peak(xt.grad[idx].cpu().numpy(), gxo)
    eg = linf_peak(gp[idx].cpu().numpy(), gpo)
    record("reverb_config4_full_size_sampled_items", y=ey.max(), gx=egx.max(), gctl=eg.max())
    assert ey.max() < 3e-5 and egx.max() < 3e-5 and eg.max() < CTL_TOL, (ey, egx, eg)


def test_reverb_unsupported_sizes_raise(D):
    """Filters longer than the 4096-point filter-bank window and impulse responses beyond 2^20 samples are refused, not approximated."""
    from dasp_pytorch_amd._lib import DaspHipError
    x = torch.rand(1, 2, 1000, device="cuda:0")
    cols = [torch.rand(1, device="cuda:0") for _ in range(25)]
    with pytest.raises(DaspHipError):
        D.noise_shaped_reverberation(x, SR, *cols, num_samples=4096, num_bandpass_taps=3587)
    with pytest.raises(DaspHipError):
        D.noise_shaped_reverberation(x, SR, *cols, num_samples=(1 << 20) + 1, num_bandpass_taps=63, noise=torch.zeros(2, 12, 8, device="cuda:0"))


def test_reverb_semantics(D):
    B, N = 2, 3000
    x = torch.rand(B, 2, N, device="cuda:0") * 2 - 1
    x0 = x.clone()
    cols = [torch.rand(B, device="cuda:0") for _ in range(24)]
    zero, one = torch.zeros(B, device="cuda:0"), torch.ones(B, device="cuda:0")
    # mix = 0 -> dry signal, input not mutated; device_noise path runs; asserts as the reference
    y = D.noise_shaped_reverberation(x, SR, *cols, zero, num_samples=512, num_bandpass_taps=31)
    assert torch.equal(y, x) and torch.equal(x, x0)
    yd = D.noise_shaped_reverberation(x, SR, *cols, one, num_samples=512, num_bandpass_taps=31, device_noise=True)
    assert yd.shape == (B, 2, N) and torch.isfinite(yd).all()
    with pytest.raises(AssertionError):
        D.noise_shaped_reverberation(x, SR, *cols, one, num_bandpass_taps=32)
    with pytest.raises(AssertionError):
        D.noise_shaped_reverberation(torch.zeros(1, 3, 100, device="cuda:0"), SR, *[c[:1] for c in cols], one[:1])
    # wet path is causal and linear in x: an impulse at n0 reproduces mix * ir delayed by n0
    imp = torch.zeros(1, 2, 2000, device="cuda:0"); imp[:, :, 100] = 1.0
    nz = torch.randn(2, 12, 512 + 30, device="cuda:0")
    c1 = [c[:1] for c in cols]
    y1 = D.noise_shaped_reverberation(imp, SR, *c1, one[:1], num_samples=512, num_bandpass_taps=31, noise=nz)
    assert y1[..., :100].abs().max().item() < 1e-6 and y1[..., 100 + 512:].abs().max().item() < 1e-6
    y2 = D.noise_shaped_reverberation(2.5 * imp, SR, *c1, one[:1], num_samples=512, num_bandpass_taps=31, noise=nz)
    assert torch.allclose(y2, 2.5 * y1, rtol=1e-4, atol=1e-6)


def test_band_split_filter_bank_equals_one_workgroup_per_window(D, monkeypatch):
    """Few batch items: the filter-bank kernel deals the 12 bands out to several workgroups per (item, window) (float atomics into the
    impulse responses; csrc/reverb.hip fb_fused_kernel) - same results as one workgroup per window, forward and every gradient."""
    rng = np.random.default_rng(5)
    B, C, N, L, taps = 2, 2, 20000, 9000, 255
    x = (rng.random((B, C, N)) * 2 - 1).astype(np.float32)
    w = rng.standard_normal((B, 2, N)).astype(np.float32)
    p = rng.random((B, 25)).astype(np.float32)
    noise = rng.standard_normal((2 * B, 12, L + taps - 1)).astype(np.float32)
    with reverb_plan(band_split=1):
        y1, gx1, gp1 = run(D, x, p, w, noise, L, taps)
    for split in (4, 12):
        with reverb_plan(band_split=split):
            y2, gx2, gp2 = run(D, x, p, w, noise, L, taps)
        assert np.abs(y2 - y1).max() <= 2e-6 * np.abs(y1).max() and np.abs(gx2 - gx1).max() <= 2e-6 * np.abs(gx1).max()
        assert np.abs(gp2 - gp1).max() <= 1e-5 * np.abs(gp1).max()


def test_envelope_inside_the_transform_equals_per_band_route(D, monkeypatch):
    """The filter bank applies the decay envelope inside the transform (weighted noise and filters, one inverse transform per window;
    csrc/reverb.hip fb_fused_kernel ROUTE 1) while |rho| 4096 <= 2 and band by band in the time domain beyond. L = 16384 puts the limit at
    decay ~ 0.7: item 0 (decays <= 0.5) takes the first route, item 1 (one band at 0.95) the second, item 2 sits on neither edge.
    Both against the oracle, and against each other with every item forced down the per-band route."""
    rng = np.random.default_rng(11)
    B, C, N, L, taps = 3, 2, 30000, 16384, 1023
    x = (rng.random((B, C, N)) * 2 - 1).astype(np.float32)
    w = rng.standard_normal((B, 2, N)).astype(np.float32)
    p = rng.random((B, 25)).astype(np.float32)
    p[0, 12:24] *= 0.5
    p[1, 12:24] *= 0.5; p[1, 17] = 0.95
    noise = rng.standard_normal((2 * B, 12, L + taps - 1)).astype(np.float32)
    with reverb_plan(band_split=1):                              # no float atomics: an item that keeps its route is bit-identical
        y1, gx1, gp1 = run(D, x, p, w, noise, L, taps)
    with reverb_plan(band_split=1, weight_limit=0.0):            # every item the per-band way
        y0, gx0, gp0 = run(D, x, p, w, noise, L, taps)
    pd = p.astype(np.float64)
    yo = orc.noise_shaped_reverberation(x, SR, pd[:, :12], pd[:, 12:24], pd[:, 24], noise, L, taps)
    gxo, gg, gd, gm = orc.noise_shaped_reverberation_vjp(x, SR, pd[:, :12], pd[:, 12:24], pd[:, 24], noise, w, L, taps)
    gpo = np.concatenate([gg, gd, gm[:, None]], 1)
    for y, gx, gp in ((y1, gx1, gp1), (y0, gx0, gp0)):
        assert np.abs(y - yo).max() <= 3e-5 * np.abs(yo).max() and np.abs(gx - gxo).max() <= 3e-5 * np.abs(gxo).max()
        assert (np.abs(gp - gpo).max(1) <= 1e-4 * np.abs(gpo).max(1)).all()
    assert np.abs(y1 - y0).max() <= 1e-5 * np.abs(y0).max() and np.abs(gx1 - gx0).max() <= 1e-5 * np.abs(gx0).max()
    assert (np.abs(gp1 - gp0).max(1) <= 3e-5 * np.abs(gp0).max(1)).all()
    assert not np.array_equal(y1[0], y0[0]) and np.array_equal(y1[1], y0[1])          # item 0 changed route, item 1 did not


def test_generated_noise_equals_the_explicit_noise_path(D):
    """device_noise=True: the noise is generated inside the filter-bank kernels (counter-based stream keyed by the seed; csrc/reverb.hip).
    dasp_reverb_noise writes that stream out; (i) it is the stream oracle/noise_stream.py specifies, (ii) fed through the explicit-noise
    path it reproduces the seeded call - output and every gradient - to rounding, on both filter-bank routes and with the bands dealt out;
    (iii) the seeded call agrees with the oracle on that noise; (iv) the seed comes from torch's global CPU generator unless given."""
    from dasp_pytorch_amd import ops
    from oracle import noise_stream as ns
    rng = np.random.default_rng(3)
    for B, C, N, L, taps, seed in ((2, 2, 9000, 5000, 255, 123456789012345), (3, 1, 30000, 16384, 1023, 7), (1, 2, 40000, 65536, 1023, 2 ** 62 + 5)):
        x = (rng.random((B, C, N)) * 2 - 1).astype(np.float32)
        w = rng.standard_normal((B, 2, N)).astype(np.float32)
        p = rng.random((B, 25)).astype(np.float32)
        if L == 16384:
            p[1, 17] = 0.95                       # item 1 takes the per-band route (see test_envelope_inside_the_transform...)
        nz = ops.reverb_noise(seed, B, 12, L + taps - 1, "cuda:0")
        e_model = np.abs(nz.cpu().numpy() - ns.noise(seed, B, 12, L + taps - 1)).max()
        assert e_model < 2e-5, e_model            # fp32 log2 / sqrt / sin / cos intrinsics against the fp64 model, values up to 4.9
        outs = []
        for kw in (dict(noise=nz), dict(device_noise=True, noise_seed=seed)):
            xt = dev(x).requires_grad_(True)
            cols = [dev(p[:, i]).requires_grad_(True) for i in range(25)]
            y = D.noise_shaped_reverberation(xt, SR, *cols, num_samples=L, num_bandpass_taps=taps, **kw)
            (y * dev(w)).sum().backward()
            outs.append((y.detach().cpu().numpy(), xt.grad.cpu().numpy(), torch.stack([c.grad for c in cols], 1).cpu().numpy()))
        (y0, gx0, gp0), (y1, gx1, gp1) = outs
        ey, egx, egp = np.abs(y1 - y0).max() / np.abs(y0).max(), np.abs(gx1 - gx0).max() / np.abs(gx0).max(), np.abs(gp1 - gp0).max() / np.abs(gp0).max()
        record(f"reverb_generated_noise[{B},{C},{N},{L},{taps}]", stream_vs_model=e_model, y=ey, gx=egx, gctl=egp)
        assert ey < 2e-6 and egx < 2e-6 and egp < 1e-5, (ey, egx, egp)
        if N <= 30000:
            pd = p.astype(np.float64)
            yo = orc.noise_shaped_reverberation(x, SR, pd[:, :12], pd[:, 12:24], pd[:, 24], nz.cpu().numpy(), L, taps)
            assert np.abs(y1 - yo).max() < 3e-5 * np.abs(yo).max()
    # seeding: torch.manual_seed makes the default draw reproducible, successive calls differ (as with the reference's CPU randn)
    xs = torch.rand(1, 2, 4000, device="cuda:0")
    cols = [torch.rand(1, device="cuda:0") for _ in range(24)] + [torch.ones(1, device="cuda:0")]
    run = lambda: D.noise_shaped_reverberation(xs, SR, *cols, num_samples=2048, num_bandpass_taps=127, device_noise=True)
    # the per-replay offset word: seed + offset, read by the kernels when they run (what a captured launch varies between replays)
    off = torch.tensor([5], dtype=torch.int64, device="cuda:0")
    n_a = ops.reverb_noise(100, 1, 12, 300, "cuda:0", seed_offset=off)
    assert torch.equal(n_a, ops.reverb_noise(105, 1, 12, 300, "cuda:0"))
    run_off = lambda: D.noise_shaped_reverberation(xs, SR, *cols, num_samples=2048, num_bandpass_taps=127, device_noise=True, noise_seed=100, noise_seed_offset=off)
    r5 = run_off(); off.add_(1); r6 = run_off()
    r6b = D.noise_shaped_reverberation(xs, SR, *cols, num_samples=2048, num_bandpass_taps=127, device_noise=True, noise_seed=106)
    assert float((r6 - r6b).abs().max()) < 1e-5 * float(r6.abs().max()) and float((r5 - r6).abs().max()) > 1e-2 * float(r5.abs().max())
    torch.manual_seed(99); a1 = run(); a2 = run()
    torch.manual_seed(99); b1 = run()
    # (one item: the bands are dealt out over workgroups that add into the impulse response with float atomics - equal to rounding, not to the bit)
    assert float((a1 - b1).abs().max()) < 1e-5 * float(a1.abs().max()) and float((a1 - a2).abs().max()) > 1e-2 * float(a1.abs().max())


@pytest.mark.parametrize("C", [2, 1])
def test_chunked_passes_equal_one_pass(D, monkeypatch, C):
    """dasp_reverb_plan(chunk = ..) (a developer argument kept for the measurement in DESIGN 3.3): the long-convolution pipeline run in passes over 2 and
    4 signals with chunk-sized scratch buffers - every per-chunk pointer offset (A per item for mono input, the items' paired spectra, the
    mix partial sums) - gives what the single pass gives, and that is held to the oracle."""
    B, N, L, taps = 5, 30000, 9000, 255
    rng = np.random.default_rng(17 + C)
    x = (rng.random((B, C, N)) * 2 - 1).astype(np.float32)
    w = rng.standard_normal((B, 2, N)).astype(np.float32)
    p = rng.random((B, 25)).astype(np.float32)
    noise = rng.standard_normal((2 * B, 12, L + taps - 1)).astype(np.float32)
    y0, gx0, gp0 = run(D, x, p, w, noise, L, taps)
    pd = p.astype(np.float64)
    yo = orc.noise_shaped_reverberation(x, SR, pd[:, :12], pd[:, 12:24], pd[:, 24], noise, L, taps)
    assert np.abs(y0 - yo).max() < 3e-5 * np.abs(yo).max()
    for chunk in (2, 4):
        with reverb_plan(chunk=chunk):
            y1, gx1, gp1 = run(D, x, p, w, noise, L, taps)
        assert np.abs(y1 - y0).max() <= 1e-6 * np.abs(y0).max()
        assert np.abs(gx1 - gx0).max() <= 1e-6 * np.abs(gx0).max()
        assert np.abs(gp1 - gp0).max() <= 2e-6 * np.abs(gp0).max()


