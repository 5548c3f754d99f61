"""torch.ops.dasp.* (csrc/torch_ext/dasp_torch_ops.cpp: TORCH_LIBRARY schemas + C++ torch::autograd::Function over the C ABI) for the four ops
of the reference's chain (examples/style_transfer.py:150-154):

* the chain pinned to the REFERENCE's golden through both bindings (torch ops and ctypes), which must also agree with each other to the bit -
  they launch the same kernels with the same arguments;
* torch.library.opcheck on every public op (schema, autograd registration, fake tensors, AOT dispatch with dynamic shapes);
* a torch.nn.Module made of the ops compiled with fullgraph=True (no graph break), gradients equal to eager;
* the reference's error behaviour as RuntimeError (TORCH_CHECK), double backward refused."""
import numpy as np
import pytest
import torch

from dasp_pytorch_amd import config

from tests.util import linf_peak, load_golden, record

pytestmark = pytest.mark.gpu
SR = 44100
DEV = "cuda:0"


@pytest.fixture(scope="module")
def T():
    assert torch.cuda.is_available()
    from dasp_pytorch_amd import _torch_ops
    assert _torch_ops.load(), "csrc/libdasp_torch.so missing or not loadable: __graft_entry__.build() builds it"
    return _torch_ops


def _chain_inputs(B=3, N=20000, seed=0, C=1):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.rand(B, C, N, device=DEV, generator=g) * 2 - 1
    ps = [torch.rand(B, n, device=DEV, generator=g).clamp(0.02, 0.98) for n in (18, 6, 25, 1)]
    w = torch.randn(B, 2, N, device=DEV, generator=g)
    return x, ps, w


@pytest.mark.parametrize("C", [1, 2])
def test_both_bindings_give_the_same_bits(T, monkeypatch, C):
    """StyleTransferChain through torch.ops.dasp.* and through the ctypes autograd.Functions: same kernels, same arguments - y, grad x and
    parameter gradients equal to the order in which fp32 atomics land (few items: the filter bank's bands are dealt out over workgroups
    that add into the impulse response, and the control gradients are sums of per-workgroup partials)."""
    from dasp_pytorch_amd.chain import StyleTransferChain
    x, ps, w = _chain_inputs(C=C)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setattr(config.plan, "torch_ops", flag != "0")
        assert T.enabled() == (flag == "1")
        chain = StyleTransferChain(SR, num_samples=8192, device_noise=True, noise_seed=11)
        xx = x.clone().requires_grad_(True)
        pp = [p.clone().requires_grad_(True) for p in ps]
        y = chain.process_normalized(xx, *pp)
        (y * w).sum().backward()
        outs.append([y.detach(), xx.grad] + [p.grad for p in pp])
    errs = [float((a - b).abs().max()) / float(b.abs().max()) for a, b in zip(*outs)]
    record(f"torch_ops_vs_ctypes_binding[C={C}]", y=errs[0], gx=errs[1], gparams=errs[2:])
    assert errs[0] <= 2e-6 and errs[1] <= 5e-6 and max(errs[2:]) <= 2e-5, errs


@pytest.mark.parametrize("binding", ["torch_ops", "ctypes"])
def test_chain_against_the_reference_through_each_binding(T, monkeypatch, binding):
    """tests/golden/chain_b2c1_n20000.npz (the reference's own EQ -> compressor -> reverb -> gain run with gradients) through each binding."""
    from dasp_pytorch_amd.chain import StyleTransferChain
    monkeypatch.setattr(config.plan, "torch_ops", binding == "torch_ops")
    g = load_golden("chain_b2c1_n20000")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    x = dev(g["x"]).requires_grad_(True)
    pp = [dev(g[k]).requires_grad_(True) for k in ("pn_eq", "pn_comp", "pn_rev", "pn_gain")]
    torch.manual_seed(int(g["noise_seed"]))
    y = StyleTransferChain(SR).process_normalized(x, *pp)
    (y * dev(g["w"])).sum().backward()
    y = y.detach().cpu().numpy()
    if linf_peak(y, g["y64"]).max() > 1e-3:
        pytest.skip("this torch build's CPU generator does not reproduce the golden's noise stream")
    errs = {"y64": linf_peak(y, g["y64"]).max(), "gx64": linf_peak(x.grad.cpu().numpy(), g["gx64"]).max()}
    for key, p in zip(("eq", "comp", "rev", "gain"), pp):
        ref = g[f"gpn_{key}64"]
        errs[f"gpn_{key}64"] = np.abs(p.grad.cpu().numpy() - ref).max() / np.abs(ref).max()
    record(f"chain_vs_reference_binding[{binding}]", **errs)
    assert errs["y64"] < 1e-5 and errs["gx64"] < 2e-5 and all(errs[f"gpn_{k}64"] < 1e-4 for k in ("eq", "comp", "rev", "gain")), errs


def _op_samples():
    from dasp_pytorch_amd import functional as F, ops
    from dasp_pytorch_amd.chain import ChainModule
    m = ChainModule(SR, num_samples=2048, num_bandpass_taps=127, noise_seed=5, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    r = lambda *s: torch.rand(*s, device=DEV, generator=g)
    B, N = 2, 6000
    x = (r(B, 2, N) * 2 - 1).requires_grad_(True)
    pn = (r(B, 18) * 0.9 + 0.05).requires_grad_(True)
    ctl = torch.stack([-40 + 30 * r(B), 1 + 8 * r(B), 5 + 50 * r(B), 1 + 8 * r(B), 6 * r(B)], 1).requires_grad_(True)
    cp, rp, gp = [(r(B, n) * 0.9 + 0.05).requires_grad_(True) for n in (6, 25, 1)]
    gains, decays, mix = r(B, 12).requires_grad_(True), r(B, 12).requires_grad_(True), r(B).requires_grad_(True)
    noise = torch.randn(2 * B, 12, 2048 + 127 - 1, device=DEV, generator=g)
    return m, {
        "parametric_eq_norm": (x, pn, float(SR), m.types, m.eq_lo, m.eq_span),
        "parametric_eq_norm_shared": (x, pn[:1].detach().requires_grad_(True), float(SR), m.types, m.eq_lo, m.eq_span),
        "dynamics_ctl": (x, ctl, 0, float(SR), 1e-8, 0),
        "dynamics_ctl_lookahead_expander": (x, ctl, 1, float(SR), 1e-8, 5),
        "chain_controls": (cp, rp, gp, m.lo, m.span),
        "reverb_generated_noise": (x, None, m.fspec, gains, decays, mix, 2048, 127, 12, 5, m.seed_offset, 0.0),
        "reverb_explicit_noise_mono": (x[:, :1].detach().requires_grad_(True), noise, m.fspec, gains, decays, mix, 2048, 127, 12, 0, None, 0.0),
    }


@pytest.mark.parametrize("case", ["parametric_eq_norm", "parametric_eq_norm_shared", "dynamics_ctl", "dynamics_ctl_lookahead_expander", "chain_controls",
                                  "reverb_generated_noise", "reverb_explicit_noise_mono"])
def test_opcheck(T, case):
    """torch.library.opcheck: the schema matches what the kernels do to their arguments, an autograd kernel is registered, the fake
    implementations give the real shapes / dtypes / devices, and AOTAutograd traces forward and backward (dynamic shapes) to the same numbers."""
    _, samples = _op_samples()
    op = getattr(torch.ops.dasp, case.split("_shared")[0].split("_lookahead")[0].split("_generated")[0].split("_explicit")[0]).default
    res = torch.library.opcheck(op, samples[case], raise_exception=True)
    assert all(v == "SUCCESS" for v in res.values()), res


@pytest.mark.parametrize("backend", ["aot_eager", "inductor"])
def test_chain_module_compiles_without_graph_breaks(T, backend):
    """chain.ChainModule (chain_controls -> parametric_eq_norm -> dynamics_ctl -> reverb, nothing else) under torch.compile(fullgraph=True):
    a graph break would raise. Output and all parameter gradients equal the eager module's."""
    import torch._dynamo
    from dasp_pytorch_amd.chain import ChainModule
    torch._dynamo.reset()
    m = ChainModule(SR, num_samples=4096, noise_seed=9, device=DEV)
    x, ps, w = _chain_inputs(B=2, N=16384, seed=4)
    outs = []
    for fn in (m, torch.compile(m, fullgraph=True, backend=backend)):
        pp = [p.clone().requires_grad_(True) for p in ps]
        y = fn(x, *pp)
        (y * w).sum().backward()
        outs.append([y.detach()] + [p.grad for p in pp])
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(a.abs().max()))
    # and it is the chain: equals StyleTransferChain with the same seed
    from dasp_pytorch_amd.chain import StyleTransferChain
    ref = StyleTransferChain(SR, num_samples=4096, device_noise=True, noise_seed=9, noise_seed_offset=m.seed_offset)
    with torch.no_grad():
        assert torch.allclose(m(x, *ps), ref.process_normalized(x, *ps), rtol=1e-5, atol=1e-5 * float(outs[0][0].abs().max()))


def test_errors_and_double_backward(T):
    d = torch.ops.dasp
    m, s = _op_samples()
    x = s["dynamics_ctl"][0]
    with pytest.raises(RuntimeError, match="must match the size"):                      # one control row per item (functional.py:330-336)
        d.dynamics_ctl(x, torch.zeros(1, 5, device=DEV), 0, float(SR), 1e-8, 0)
    with pytest.raises(RuntimeError, match="ROCm device"):
        d.dynamics_ctl(x.cpu(), torch.zeros(2, 5), 0, float(SR), 1e-8, 0)
    with pytest.raises(RuntimeError, match="float32"):
        d.parametric_eq_norm(x.double(), s["parametric_eq_norm"][1], float(SR), m.types, m.eq_lo, m.eq_span)
    with pytest.raises(RuntimeError, match="mono or stereo"):
        d.reverb(torch.zeros(2, 3, 64, device=DEV), None, m.fspec, *s["reverb_generated_noise"][3:])
    with pytest.raises(RuntimeError, match="is invalid for band gains"):                # (k, nb) stack with k != bs (functional.py:498-544)
        d.reverb(x, None, m.fspec, torch.zeros(1, 12, device=DEV), torch.zeros(1, 12, device=DEV), torch.zeros(2, device=DEV), 2048, 127, 12, 5, None, 0.0)
    # hand-written adjoints are once-differentiable: the backward ops carry no derivative formula
    xx = x.detach().clone().requires_grad_(True)
    y = d.dynamics_ctl(xx, s["dynamics_ctl"][1].detach(), 0, float(SR), 1e-8, 0)
    (gx,) = torch.autograd.grad(y.sum(), xx, create_graph=True)
    assert gx.requires_grad
    with pytest.raises(RuntimeError, match="not implemented"):
        gx.sum().backward()
    # bumping the seed offset between forward and backward trips autograd's version check (the adjoint would regenerate other noise)
    off = torch.zeros(1, dtype=torch.int64, device=DEV)
    g = s["reverb_generated_noise"]
    y = d.reverb(g[0], None, m.fspec, g[3], g[4], g[5], 2048, 127, 12, 5, off, 0.0)
    off.add_(1)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        y.sum().backward()


def test_inference_and_empty(T):
    d = torch.ops.dasp
    m, s = _op_samples()
    with torch.no_grad():
        y = d.parametric_eq_norm(*s["parametric_eq_norm"])
    assert not y.requires_grad and torch.isfinite(y).all()
    e = d.dynamics_ctl(torch.zeros(0, 2, 64, device=DEV), torch.zeros(0, 5, device=DEV), 0, float(SR), 1e-8, 0)
    assert e.shape == (0, 2, 64)


# ---- round 5: torch.ops.dasp.* for the reference's own signatures ---------------------------------------------------------------------------
def _ref_signature_samples():
    from dasp_pytorch_amd import functional as F
    g = torch.Generator(device=DEV).manual_seed(21)
    r = lambda *s: torch.rand(*s, device=DEV, generator=g)
    B, C, N = 3, 2, 20000
    x = (r(B, C, N) * 2 - 1).requires_grad_(True)
    from bench import PEQ_RANGES
    eq = [(r(B) * (hi - lo) + lo).requires_grad_(True) for lo, hi in PEQ_RANGES]
    comp = [(-40 + 30 * r(B)).requires_grad_(True), (1 + 8 * r(B)).requires_grad_(True), (5 + 50 * r(B)).requires_grad_(True),
            (5 + 50 * r(B)).requires_grad_(True), (1 + 8 * r(B)).requires_grad_(True), (6 * r(B)).requires_grad_(True)]
    rev = [r(B).requires_grad_(True) for _ in range(25)]
    w = torch.randn(B, C, N, device=DEV, generator=g)
    sos = torch.stack([torch.tensor([[1.0, -1.2, 0.5, 1.0, -1.5, 0.7], [0.8, 0.1, 0.2, 1.0, -0.3, 0.4], [1.1, 0.0, -0.2, 2.0, 0.4, 0.1]])] * B).to(DEV)
    return F, x, eq, comp, rev, w, sos.requires_grad_(True)


def _grads(y, w, leaves):
    for t in leaves:
        t.grad = None
    (y * w).sum().backward()
    return [y.detach().clone()] + [t.grad.clone() if t.grad is not None else None for t in leaves]


@pytest.mark.parametrize("op", ["parametric_eq", "compressor", "expander", "gain", "distortion", "sosfilt", "reverb"])
def test_reference_signatures_through_both_bindings(T, monkeypatch, op):
    """functional.* / signal.sosfilt_via_fsm on the reference's own argument lists go through torch.ops.dasp.* (round 5: parametric_eq on its 18
    control tensors, dynamics on six, gain, distortion, sosfilt, noise_shaped_reverb on 25) - same kernels and arguments as the ctypes
    autograd.Functions: outputs, input gradients and every control gradient agree to the order fp32 atomics land in."""
    import dasp_pytorch_amd as D
    F, x, eq, comp, rev, w, sos = _ref_signature_samples()
    calls = {
        "parametric_eq": (lambda: F.parametric_eq(x, SR, *eq), [x] + eq, "parametric_eq"),
        "compressor": (lambda: F.compressor(x, SR, *comp, lookahead_samples=3), [x] + comp, "dynamics"),
        "expander": (lambda: F.expander(x, SR, *comp), [x] + comp, "dynamics"),
        "gain": (lambda: F.gain(x, SR, eq[0]), [x, eq[0]], "gain"),
        "distortion": (lambda: F.distortion(x, SR, torch.stack([comp[5], comp[5]], 1)), [x, comp[5]], "distortion"),
        "sosfilt": (lambda: D.signal.sosfilt_via_fsm(sos, x), [x, sos], "sosfilt"),
        "reverb": (lambda: F.noise_shaped_reverberation(x, SR, *rev, num_samples=4096, num_bandpass_taps=255, noise_seed=17), [x] + rev, "noise_shaped_reverb"),
    }
    fn, leaves, opname = calls[op]
    seen = []
    real = getattr(torch.ops.dasp, opname)

    class Spy:                                  # counts the calls that reach the registered op
        def __getattr__(self, name):
            return getattr(real, name)

        def __call__(self, *a, **k):
            seen.append(opname)
            return real(*a, **k)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setattr(config.plan, "torch_ops", flag != "0")
        if flag == "1":
            monkeypatch.setattr(torch.ops.dasp, opname, Spy(), raising=False)
        y = fn()
        outs.append(_grads(y, w, leaves))
        if flag == "1":
            monkeypatch.undo()
            assert seen, f"functional.{op} did not reach torch.ops.dasp.{opname}"
    errs = []
    for a, b in zip(*outs):
        assert (a is None) == (b is None)
        if a is not None:
            errs.append(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30))
    record(f"reference_signature_torch_ops_vs_ctypes[{op}]", y=errs[0], gx=errs[1], gcontrols=max(errs[2:]))
    assert errs[0] <= 2e-6 and errs[1] <= 5e-6 and max(errs[2:]) <= 2e-5, errs
    if op in ("compressor", "expander"):        # release_ms: accepted, zero gradient (functional.py:340,343-344)
        assert float(outs[0][1 + 4].abs().max()) == 0.0


def _new_op_samples():
    F, x, eq, comp, rev, w, sos = _ref_signature_samples()
    from dasp_pytorch_amd import ops
    from dasp_pytorch_amd.functional import _PEQ_TYPES, _device_filterbank
    filters = _device_filterbank(255, float(SR), torch.device(DEV))
    xs = x[:2, :, :6000].detach().clone().requires_grad_(True)
    cut = lambda ts: [t[:2].detach().clone().requires_grad_(True) for t in ts]
    fspec = ops._reverb_fspec(xs, filters, 2048)
    return {
        "parametric_eq": (xs, float(SR), cut(eq), list(_PEQ_TYPES)),
        "parametric_eq_shared": (xs, float(SR), [t[:1].detach().clone().requires_grad_(True) for t in eq], list(_PEQ_TYPES)),
        "dynamics": (xs, float(SR), *cut(comp), 1e-8, 0, 0),
        "dynamics_lookahead_expander": (xs, float(SR), *cut(comp), 1e-8, 4, 1),
        "gain": (xs, cut([eq[0]])[0]),
        "distortion": (xs, torch.rand(2, 2, device=DEV).requires_grad_(True)),
        "sosfilt": (sos[:2].detach().clone().requires_grad_(True), xs),
        "sosfilt_shared": (sos[:1].detach().clone().requires_grad_(True), xs),
        "noise_shaped_reverb": (xs, cut(rev[:12]), cut(rev[12:24]), cut(rev[24:])[0], None, fspec, 2048, 255, 5, None, 0.0),
    }


@pytest.mark.parametrize("case", ["parametric_eq", "parametric_eq_shared", "dynamics", "dynamics_lookahead_expander", "gain", "distortion", "sosfilt",
                                  "sosfilt_shared", "noise_shaped_reverb"])
def test_opcheck_reference_signatures(T, case):
    samples = _new_op_samples()
    op = getattr(torch.ops.dasp, case.split("_shared")[0].split("_lookahead")[0]).default
    if case == "noise_shaped_reverb":
        # few items: the filter bank deals its bands out over workgroups that add into the impulse responses with float atomics - the order
        # is not fixed, and opcheck compares eager gradients with AOTDispatcher's at a tight tolerance (one run in ~10 differed); one
        # workgroup per window makes the op deterministic for the comparison
        from tests.test_gpu_reverb import reverb_plan
        with reverb_plan(band_split=1):
            res = torch.library.opcheck(op, samples[case], raise_exception=True)
    else:
        res = torch.library.opcheck(op, samples[case], raise_exception=True)
    assert all(v == "SUCCESS" for v in res.values()), res


@pytest.mark.parametrize("backend", ["aot_eager", "inductor"])
def test_functional_calls_compile_without_graph_breaks(T, backend):
    """A module that calls functional.parametric_eq and functional.compressor directly - the reference's signatures - under
    torch.compile(fullgraph=True): the functions resolve to torch.ops.dasp.parametric_eq / dynamics (round 4: every direct functional.*
    call was a graph break). Output and all gradients equal the eager module's."""
    import torch._dynamo
    import dasp_pytorch_amd as D
    torch._dynamo.reset()

    class Net(torch.nn.Module):
        def forward(self, x, eq, comp):
            y = D.functional.parametric_eq(x, SR, *eq)
            return D.functional.compressor(y, SR, *comp)
    F, x, eq, comp, rev, w, sos = _ref_signature_samples()
    m = Net()
    outs = []
    for fn in (m, torch.compile(m, fullgraph=True, backend=backend)):
        leaves = [x] + eq + comp
        y = fn(x, eq, comp)
        outs.append(_grads(y, w, leaves))
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(a.abs().max()) + 1e-12)


def test_range_flag_words(T):
    """The normalised ops' `range_flag`: the design kernel / the chain's control kernel OR bit i into the device word when column i leaves
    [0, 1] (NaN passes, as in the reference); the words are sticky; modules._FlagRangeCheck turns them into the reference's ValueError."""
    d = torch.ops.dasp
    m, s = _op_samples()
    x, pn = s["parametric_eq_norm"][0].detach(), s["parametric_eq_norm"][1].detach().clone()
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    d.parametric_eq_norm(x, pn, float(SR), m.types, m.eq_lo, m.eq_span, flag)
    assert int(flag) == 0
    pn[1, 4] = 1.25; pn[0, 17] = -0.1; pn[0, 2] = float("nan")
    d.parametric_eq_norm(x, pn, float(SR), m.types, m.eq_lo, m.eq_span, flag)
    assert int(flag) == (1 << 4) | (1 << 17)
    cp, rp, gp = (t.detach().clone() for t in s["chain_controls"][:3])
    f2 = torch.zeros(1, dtype=torch.int32, device=DEV)
    d.chain_controls(cp, rp, gp, m.lo, m.span, f2)
    assert int(f2) == 0
    cp[0, 3] = 2.0; rp[1, 13] = -0.5; gp[1, 0] = 1.5                      # release_ms (read by nothing, checked like the others), band1_decay, gain_db
    d.chain_controls(cp, rp, gp, m.lo, m.span, f2)
    assert int(f2) & 0xFFFFFFFF == (1 << 3) | (1 << (6 + 13)) | (1 << 31)
    d.chain_controls(s["chain_controls"][0].detach(), s["chain_controls"][1].detach(), s["chain_controls"][2].detach(), m.lo, m.span, f2)
    assert int(f2) & 0xFFFFFFFF == (1 << 3) | (1 << (6 + 13)) | (1 << 31)    # sticky: nobody cleared it
