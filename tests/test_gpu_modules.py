"""Processor wrappers on the GPU: process_normalized equals the functional call with de-normalised controls, gradients
reach the normalised parameter tensor, and the reference's EQ -> compressor -> reverb -> gain chain
(examples/style_transfer.py:150-154) runs end to end with finite gradients."""
import pytest
import torch

from dasp_pytorch_amd import config

pytestmark = pytest.mark.gpu
SR = 44100


@pytest.fixture(scope="module")
def D():
    assert torch.cuda.is_available()
    import dasp_pytorch_amd as D
    return D


def test_process_normalized_matches_functional(D):
    g = torch.Generator(device="cuda:0").manual_seed(0)
    B, N = 4, 8192
    x = torch.rand(B, 2, N, device="cuda:0", generator=g) * 2 - 1
    eq = D.ParametricEQ(SR)
    p = torch.rand(B, 18, device="cuda:0", generator=g, requires_grad=True)
    y = eq.process_normalized(x, p)
    lo = torch.tensor([r[0] for r in eq.param_ranges.values()], device="cuda:0")
    hi = torch.tensor([r[1] for r in eq.param_ranges.values()], device="cuda:0")
    d = p.detach() * (hi - lo) + lo
    # (the fused op de-normalises in fp64 inside the design kernel; the functional call gets the fp32-rounded physical values)
    assert torch.allclose(y, D.parametric_eq(x, SR, *[d[:, i] for i in range(18)]), rtol=0, atol=2e-5 * float(y.detach().abs().max()))
    y.square().mean().backward()
    assert p.grad.shape == p.shape and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0
    with pytest.raises(ValueError, match="band2_gain_db"):
        bad = p.detach().clone(); bad[0, 9] = -0.1
        eq.process_normalized(x, bad)
    assert torch.isfinite(eq.process_normalized(x, p.detach())).all()      # a refused call leaves no state behind
    eq.validate_range = False                                              # opt-out: no read-back, out-of-range values are simply used
    assert torch.isfinite(eq.process_normalized(x, bad)).all()
    eq.validate_range = True
    # Compressor / reverb hand their de-normalised matrix to the kernels as one tensor: same numbers as the per-control functional calls
    comp = D.Compressor(SR)
    pc = torch.rand(B, 6, device="cuda:0", generator=g, requires_grad=True)
    yc = comp.process_normalized(x, pc)
    lo = torch.tensor([r[0] for r in comp.param_ranges.values()], device="cuda:0"); hi = torch.tensor([r[1] for r in comp.param_ranges.values()], device="cuda:0")
    dc = (pc.detach() * (hi - lo) + lo).requires_grad_(True)
    yc2 = D.compressor(x, SR, *[dc[:, i] for i in range(6)])
    assert torch.equal(yc, yc2)
    w = torch.randn(B, 2, N, device="cuda:0", generator=g)
    (yc * w).sum().backward(); (yc2 * w).sum().backward()
    assert torch.allclose(pc.grad, dc.grad * (hi - lo), rtol=1e-5, atol=1e-9) and float(pc.grad[:, 3].abs().max()) == 0.0
    with pytest.raises(ValueError, match="knee_db"):
        badc = pc.detach().clone(); badc[1, 4] = 1.5
        comp.process_normalized(x, badc)
    # Distortion works here (it raises in the reference); mono so that (bs,) drives are legal
    dist = D.Distortion()
    assert torch.isfinite(dist.process_normalized(x[:, :1], torch.rand(B, 1, device="cuda:0", generator=g))).all()


def _norm_golden(D, name, mod, **kw):
    import numpy as np
    from tests.util import linf_peak, load_golden
    g = load_golden(name)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    x = dev(g["x"]).requires_grad_(True)
    pn = dev(g["pn"]).requires_grad_(True)
    if "noise_seed" in g:
        torch.manual_seed(int(g["noise_seed"]))          # the reference's noise comes from the global CPU generator (functional.py:548)
    y = mod.process_normalized(x, pn, **kw)
    (y * dev(g["w"])).sum().backward()
    return g, y.detach().cpu().numpy(), x.grad.cpu().numpy(), pn.grad.cpu().numpy(), linf_peak


def test_process_normalized_against_reference_goldens(D):
    """Processor.process_normalized (dasp_pytorch/modules.py:25-51, 70-91) on the reference's own outputs: the normalised (bs, P)
    tensor goes in, y / grad x / the gradient w.r.t. the normalised parameters come out - forward 1e-5 and gradients 1e-4 against the
    reference's fp64 run, and inside the literal 1e-4 bar against its fp32 output (tests/golden/make_golden.py norm_case)."""
    for name, mod, tol_p in (("norm_gain_b3c2_n4000", D.Gain(SR), 1e-5), ("norm_eq_b3c2_n12000", D.ParametricEQ(SR), 1e-4),
                             ("norm_comp_b3c2_n12000", D.Compressor(SR), 1e-4), ("norm_rev_b1c2_n6000", D.NoiseShapedReverb(SR), 1e-4)):
        g, y, gx, gpn, linf_peak = _norm_golden(D, name, mod)
        if name.startswith("norm_rev") and linf_peak(y, g["y64"]).max() > 1e-3:
            pytest.skip("this torch build's CPU generator does not reproduce the golden's noise stream")
        assert linf_peak(y, g["y64"]).max() < 1e-5, name
        assert linf_peak(y, g["y32"]).max() < 1e-4, name
        assert linf_peak(gx, g["gx64"]).max() < 2e-5, name
        if name.startswith("norm_comp"):     # per column: the six control gradients differ by orders of magnitude; release_ms is exactly 0
            import numpy as np
            from tests.util import record
            cols = [np.abs(gpn[:, j] - g["gpn64"][:, j]).max() / max(np.abs(g["gpn64"][:, j]).max(), 1e-12) for j in range(6)]
            record("norm_comp_golden_param_grads", cols=cols)
            assert max(cols) < 1e-4, (name, cols)
        else:
            assert linf_peak(gpn, g["gpn64"]).max() < tol_p, name


def test_reference_style_subclass_and_live_ranges(D):
    """A processor written against the reference (sets only sample_rate / process_fn / param_ranges / num_params, modules.py:94-107)
    works unchanged, and editing param_ranges after construction changes the de-normalisation, as it does in the reference."""
    class MyGain(D.Processor):
        def __init__(self, sample_rate):
            super().__init__()
            self.sample_rate = sample_rate
            self.process_fn = D.gain
            self.param_ranges = {"gain_db": (-6.0, 6.0)}
            self.num_params = len(self.param_ranges)
    m = MyGain(SR)
    x = torch.rand(2, 1, 256, device="cuda:0")
    p = torch.tensor([[0.0], [1.0]], device="cuda:0")
    y = m.process_normalized(x, p)
    assert torch.allclose(y[0], x[0] * 10 ** (-6 / 20), rtol=1e-5) and torch.allclose(y[1], x[1] * 10 ** (6 / 20), rtol=1e-5)
    m.param_ranges["gain_db"] = (0.0, 20.0)
    y = m.process_normalized(x, p)
    assert torch.allclose(y[0], x[0], rtol=1e-6) and torch.allclose(y[1], x[1] * 10.0, rtol=1e-5)
    assert m.num_params == 1
    # positional constructor arguments in the reference's order (modules.py:110-121, 159-187)
    assert D.Distortion(3.0, 9.0).param_ranges == {"drive_db": (3.0, 9.0)}
    assert D.Compressor(SR, -40.0, -10.0).param_ranges["threshold_db"] == (-40.0, -10.0)


def test_style_transfer_chain(D):
    g = torch.Generator(device="cuda:0").manual_seed(1)
    B, N = 3, 32768
    x = torch.rand(B, 1, N, device="cuda:0", generator=g) * 2 - 1
    mods = [D.ParametricEQ(SR), D.Compressor(SR), D.NoiseShapedReverb(SR), D.Gain(SR)]
    ps = [torch.rand(B, m.num_params, device="cuda:0", generator=g).clamp(0.01, 0.99).requires_grad_(True) for m in mods]
    y = x
    for m, p in zip(mods, ps):
        y = m.process_normalized(y, p)
    assert y.shape == (B, 2, N) and torch.isfinite(y).all()
    y.square().mean().backward()
    for m, p in zip(mods, ps):
        assert torch.isfinite(p.grad).all(), type(m).__name__
    assert ps[1].grad[:, 3].abs().max() == 0          # release_ms: no path to the output


@pytest.mark.parametrize("graph", [False, True])
def test_style_transfer_chain_example_runs(graph):
    """BASELINE config 5 on synthetic clips (examples/style_transfer_synth.py): predictor -> EQ -> compressor -> reverb -> MR-STFT loss,
    backward through all three effects into every predictor parameter, optimizer steps with finite losses; eagerly and as a replayed
    HIP graph (--graph)."""
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "style_transfer_synth.py")
    spec = importlib.util.spec_from_file_location("style_transfer_synth", path)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    out, model = mod.run(steps=3, batch=2, n=32768, ir_samples=8192, width=8, quiet=True, graph=graph)
    assert out["finite"] and out["steps"] == 3
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    assert any(float(p.grad.abs().max()) > 0 for p in model.parameters())


def test_style_transfer_chain_folds_the_gain(D):
    """dasp_pytorch_amd.chain.StyleTransferChain (gain folded into the compressor's make-up gain) equals the four processors run in
    sequence, forward and for every parameter gradient: the reverb is linear per item, so the per-item gain commutes with it."""
    from dasp_pytorch_amd.chain import StyleTransferChain
    g = torch.Generator(device="cuda:0").manual_seed(5)
    B, N = 3, 24576
    x = torch.rand(B, 2, N, device="cuda:0", generator=g) * 2 - 1
    chain = StyleTransferChain(SR, num_samples=4096)
    ps = [torch.rand(B, n, device="cuda:0", generator=g).clamp(0.02, 0.98) for n in chain.num_params]
    w = torch.randn(B, 2, N, device="cuda:0", generator=g)
    outs = []
    for fused in (True, False):
        pp = [p.clone().requires_grad_(True) for p in ps]
        xx = x.clone().requires_grad_(True)
        torch.manual_seed(11)                      # same reverb noise for both runs (drawn from the CPU generator)
        if fused:
            y = chain.process_normalized(xx, *pp)
        else:
            y = chain.equalizer.process_normalized(xx, pp[0])
            y = chain.compressor.process_normalized(y, pp[1])
            y = chain.reverb.process_normalized(y, pp[2])
            y = chain.gain.process_normalized(y, pp[3])
        (y * w).sum().backward()
        outs.append((y.detach(), xx.grad, [p.grad for p in pp]))
    (y1, gx1, gp1), (y2, gx2, gp2) = outs
    peak = float(y2.abs().max())
    assert float((y1 - y2).abs().max()) < 2e-5 * peak
    assert float((gx1 - gx2).abs().max()) < 5e-5 * float(gx2.abs().max())
    from tests.util import record
    eg = [float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12) for a, b in zip(gp1, gp2)]
    record("chain_folded_vs_sequence", y=float((y1 - y2).abs().max()) / peak, gx=float((gx1 - gx2).abs().max()) / float(gx2.abs().max()), gparams=eg)
    # both sides are pinned to the reference itself by tests/test_gpu_chain.py::test_chain_with_gradients_against_the_reference;
    # this HIP-vs-HIP comparison measured 1e-6 (profiles/r04/parity_measured.jsonl)
    assert max(eg) < 2e-5, eg


def test_chain_controls_in_one_launch_equal_the_torch_ops(D, monkeypatch):
    """The chain de-normalises the compressor's, the reverb's and the gain's parameters and folds the gain in one launch per direction
    (dasp_chain_controls / ops.ChainControlsFunction); with config.plan.chain_fused_controls = False the same is done by torch ops on the tensors.
    Mono input (the reference's training shape), outputs and all 50 parameter gradients; the release_ms column gets an exact zero."""
    from dasp_pytorch_amd.chain import StyleTransferChain
    g = torch.Generator(device="cuda:0").manual_seed(9)
    B, N = 4, 20000
    x = torch.rand(B, 1, N, device="cuda:0", generator=g) * 2 - 1
    chain = StyleTransferChain(SR, num_samples=4096)
    ps = [torch.rand(B, n, device="cuda:0", generator=g).clamp(0.02, 0.98) for n in chain.num_params]
    w = torch.randn(B, 2, N, device="cuda:0", generator=g)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setattr(config.plan, "chain_fused_controls", flag != "0")
        pp = [p.clone().requires_grad_(True) for p in ps]
        torch.manual_seed(13)
        y = chain.process_normalized(x, *pp)
        (y * w).sum().backward()
        outs.append((y.detach(), [p.grad for p in pp]))
    (y1, gp1), (y0, gp0) = outs
    assert float((y1 - y0).abs().max()) <= 1e-6 * float(y0.abs().max())
    for a, b in zip(gp1, gp0):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-12)
    assert float(gp1[1][:, 3].abs().max()) == 0.0
    bad = [p.clone() for p in ps]
    bad[2][1, 7] = 1.5
    with pytest.raises(ValueError, match="band7_gain"):
        chain.process_normalized(x, *bad)


def test_parameter_rows_that_do_not_match_the_batch_are_refused(D):
    """Only the EQ broadcasts a parameter batch of 1 over the batch (functional.py:208-220); compressor, gain and
    reverb raise RuntimeError in the reference. The kernels read controls at ctl[b * n ...], so a (1, P) tensor with bs > 1 must never
    reach them - neither through the chain (fused and torch-op control paths), nor through the matrix ops, nor as mis-sized reverb columns."""
    from dasp_pytorch_amd import ops
    from dasp_pytorch_amd.chain import StyleTransferChain
    B, N = 3, 9000
    x = torch.rand(B, 2, N, device="cuda:0") * 2 - 1
    chain = StyleTransferChain(SR, num_samples=2048)
    full = [torch.rand(B, n, device="cuda:0").clamp(0.05, 0.95) for n in chain.num_params]
    assert torch.isfinite(chain.process_normalized(x, full[0][:1], *full[1:])).all()           # the EQ's broadcast is legal
    for k in (1, 2, 3):
        ps = [p.clone() for p in full]
        ps[k] = ps[k][:1]
        with pytest.raises(RuntimeError):
            chain.process_normalized(x, *ps)
    with pytest.raises(RuntimeError):
        ops.DynamicsMatrixFunction.apply(x, 0, float(SR), 1e-8, 0, torch.rand(1, 6, device="cuda:0"))
    with pytest.raises(RuntimeError):
        ops.DynamicsCtlFunction.apply(x, 0, float(SR), 1e-8, 0, torch.rand(1, 5, device="cuda:0"))
    with pytest.raises(RuntimeError):
        ops.DynamicsMatrixFunction.apply(x, 0, float(SR), 1e-8, 0, torch.rand(B, 5, device="cuda:0"))
    # reverb columns with k != bs values each, 12 k divisible by bs (the reference's .view(bs, 12) raises)
    cols = [torch.rand(2 * B, device="cuda:0") for _ in range(24)]
    with pytest.raises(RuntimeError):
        D.noise_shaped_reverberation(x, SR, *cols, torch.rand(B, device="cuda:0"), num_samples=512, num_bandpass_taps=31)
    # signal.biquad on zero filters: empty design, empty gradients of the recorded shapes
    e = torch.zeros(0, 1, 1, device="cuda:0", requires_grad=True)
    b, a = D.signal.biquad(e, e.detach().clone().requires_grad_(True), e.detach().clone().requires_grad_(True), SR, "peaking")
    (b.sum() + a.sum()).backward()
    assert e.grad is not None and e.grad.shape == e.shape


def test_hip_graph_capture_replays_the_eager_step(D):
    """The ops are plain stream launches, so a training step through them can be captured into a HIP graph (torch.cuda.CUDAGraph)
    and replayed on new data: EQ -> compressor -> gain on normalised controls plus the multi-resolution STFT loss, forward and
    backward, gives the eager step's outputs and loss bit for bit and its gradients to the last few bits."""
    g = torch.Generator(device="cuda:0").manual_seed(5)
    B, N = 4, 16384
    eq, comp, gain = D.ParametricEQ(SR), D.Compressor(SR), D.Gain(SR)
    loss_fn = D.losses.MultiResolutionSTFTLoss()
    sizes = [eq.num_params, comp.num_params, gain.num_params]

    def step(x, target, p):
        pe, pc, pg = torch.split(p, sizes, dim=1)
        y = gain.process_normalized(comp.process_normalized(eq.process_normalized(x, pe), pc), pg)
        loss = loss_fn(y, target)
        loss.backward()
        return y, loss

    def data():
        x = torch.rand(B, 2, N, device="cuda:0", generator=g) * 2 - 1
        t = torch.rand(B, 2, N, device="cuda:0", generator=g) - 0.5
        p = torch.rand(B, sum(sizes), device="cuda:0", generator=g).clamp(0.02, 0.98)
        return x, t, p

    sx, st, sp = data()
    sx.requires_grad_(True); sp.requires_grad_(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            sx.grad = sp.grad = None
            step(sx, st, sp)
    torch.cuda.current_stream().wait_stream(side)
    sx.grad = sp.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sy, sloss = step(sx, st, sp)
    for _ in range(2):                                    # replay on fresh data, compare with the eager step on the same data
        x, t, p = data()
        with torch.no_grad():
            sx.copy_(x); st.copy_(t); sp.copy_(p)
        graph.replay()
        xe, pe = x.clone().requires_grad_(True), p.clone().requires_grad_(True)
        ye, le = step(xe, t, pe)
        assert torch.equal(sy, ye) and torch.equal(sloss, le)
        # (the loss' backward kernel accumulates overlapping frames with atomics: summation order, hence the last bits, vary run to run)
        for a, b in ((sx.grad, xe.grad), (sp.grad, pe.grad)):
            assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())
        assert torch.isfinite(sp.grad).all() and sp.grad.abs().sum() > 0


def test_empty_inputs_return_empty_tensors(D):
    """An empty batch / zero-length signal returns an empty tensor (and zero control gradients), as the reference's tensor ops do,
    instead of a launch error."""
    for shape in ((0, 2, 512), (2, 2, 0)):
        x = torch.zeros(shape, device="cuda:0", requires_grad=True)
        bs = shape[0]
        g = torch.zeros(bs, device="cuda:0", requires_grad=True)
        y = D.gain(x, SR, g)
        assert y.shape == x.shape
        y.sum().backward()
        assert g.grad.shape == g.shape and float(g.grad.abs().sum()) == 0.0
        cols = [torch.ones(bs, device="cuda:0") * v for v in (0.0, 100.0, 1.0) * 6]
        assert D.parametric_eq(x, SR, *cols).shape == x.shape
        assert D.compressor(x, SR, *[torch.ones(bs, device="cuda:0")] * 6).shape == x.shape


def test_host_and_device_tensors_in_one_call_are_refused(D):
    """The one-GPU half of the device guards: a parameter tensor left on the host is refused by name (the kernels take raw pointers)."""
    from dasp_pytorch_amd._lib import DaspHipError
    x = torch.rand(2, 2, 4096, device="cuda:0")
    with pytest.raises(DaspHipError, match="cpu"):
        D.signal.sosfilt_via_fsm(torch.rand(2, 2, 6), x)
    with pytest.raises(DaspHipError, match="cpu"):
        D.losses.MultiResolutionSTFTLoss()(x, x.cpu())


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process")
def test_tensors_on_a_non_current_device(D):
    """x on cuda:1 while cuda:0 is the current device: kernels are launched on x's device and stream (ops wrap every call in
    torch.cuda.device(x.device)); tensors of different devices in one call are refused."""
    from dasp_pytorch_amd._lib import DaspHipError
    g = torch.Generator(device="cuda:1").manual_seed(0)
    x1 = torch.rand(2, 2, 8192, device="cuda:1", generator=g) * 2 - 1
    p = torch.rand(2, 18, device="cuda:1", generator=g)
    assert torch.cuda.current_device() == 0
    eq = D.ParametricEQ(SR)
    y1 = eq.process_normalized(x1, p)
    y0 = eq.process_normalized(x1.to("cuda:0"), p.to("cuda:0"))
    assert y1.device == x1.device and torch.equal(y1.cpu(), y0.cpu())
    with pytest.raises(DaspHipError):
        D.signal.sosfilt_via_fsm(torch.rand(2, 2, 6, device="cuda:0"), x1)


def test_chain_as_graphed_callable(D):
    """torch.cuda.make_graphed_callables over the chain's process_normalized: forward and backward become HIP-graph replays (no Python, no
    ctypes, no autograd bookkeeping per step - the answer to the host time of eager steps at the reference's batch sizes). The ops
    are plain stream launches without host read-backs (the [0, 1] check is skipped while capturing), so they capture as they are; the
    reverb's generated noise takes a fixed base seed plus a device offset word that the caller bumps between replays. Outputs and all 50
    parameter gradients equal the eager call's with the same seed and offset."""
    from dasp_pytorch_amd.chain import StyleTransferChain
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(31)
    B, N = 4, 32768
    off = torch.zeros(1, dtype=torch.int64, device=dev)
    chain = StyleTransferChain(SR, num_samples=4096, device_noise=True, noise_seed=1234, noise_seed_offset=off)
    x = torch.rand(B, 1, N, device=dev, generator=g) * 2 - 1
    ps = [torch.rand(B, n, device=dev, generator=g).clamp(0.05, 0.95) for n in chain.num_params]
    w = torch.randn(B, 2, N, device=dev, generator=g)
    fn = lambda x_, a, b, c, d: chain.process_normalized(x_, a, b, c, d)
    sample = (x.clone(),) + tuple(p.clone().requires_grad_(True) for p in ps)
    graphed = torch.cuda.make_graphed_callables(fn, sample)
    for step in range(3):
        off.fill_(step)
        pe = [p.clone().requires_grad_(True) for p in ps]
        ye = fn(x, *pe)
        (ye * w).sum().backward()
        pg = [p.clone().requires_grad_(True) for p in ps]
        yg = graphed(x, *pg)
        (yg * w).sum().backward()
        assert float((yg - ye).detach().abs().max()) <= 2e-6 * float(ye.detach().abs().max())
        for a, b in zip(pg, pe):
            assert float((a.grad - b.grad).abs().max()) <= 1e-5 * max(float(b.grad.abs().max()), 1e-12)
        if step:
            assert float((yg.detach() - y_prev).abs().max()) > 1e-3 * float(yg.detach().abs().max())       # the offset word changed the noise of the replay
        y_prev = yg.detach().clone()


def test_deferred_range_check_raises_one_call_late(D):
    """validate_range = "deferred": no host wait in process_normalized; a value outside [0, 1] raises the reference's ValueError (the parameter
    named) at the next call or at flush_range_check(), and a good call after a flushed bad one runs."""
    g = torch.Generator(device="cuda:0").manual_seed(2)
    x = torch.rand(2, 2, 4096, device="cuda:0", generator=g) * 2 - 1
    eq = D.ParametricEQ(SR)
    eq.validate_range = "deferred"
    good = torch.rand(2, 18, device="cuda:0", generator=g)
    bad = good.clone(); bad[1, 4] = 1.25                                  # band0_cutoff_freq
    y = eq.process_normalized(x, good)
    assert torch.isfinite(y).all()
    eq.process_normalized(x, bad)                                         # queued; nothing read back yet
    with pytest.raises(ValueError, match="band0_cutoff_freq"):
        eq.process_normalized(x, good)                                    # the previous call's numbers arrive here
    assert torch.isfinite(eq.process_normalized(x, good)).all()
    eq.process_normalized(x, bad)
    with pytest.raises(ValueError, match="band0_cutoff_freq"):
        eq.flush_range_check()
    eq.flush_range_check()                                                # nothing pending: no-op
    # the chain: one deferred check for all 50 parameters
    from dasp_pytorch_amd.chain import StyleTransferChain
    chain = StyleTransferChain(SR, num_samples=2048, device_noise=True, noise_seed=3)
    for p in (chain.equalizer, chain.compressor, chain.reverb, chain.gain):
        p.validate_range = "deferred"
    ps = [torch.rand(2, n, device="cuda:0", generator=g) for n in chain.num_params]
    chain.process_normalized(x, *ps)
    ps[2][0, 13] = -0.5                                                   # band1_decay
    chain.process_normalized(x, *ps)
    with pytest.raises(ValueError, match="band1_decay"):
        chain.flush_range_check()
