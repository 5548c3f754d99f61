"""Processor wrappers (dasp_pytorch_amd.modules) against the reference's tables and semantics
(dasp_pytorch/modules.py). The expected ranges below are the reference's defaults, restated."""
import pytest
import torch

import dasp_pytorch_amd as D

SR = 44100
EXPECTED = {
    "Gain": [("gain_db", -24.0, 24.0)],
    "Compressor": [("threshold_db", -60.0, 0.0), ("ratio", 1.0, 20.0), ("attack_ms", 5.0, 100.0), ("release_ms", 5.0, 100.0),
                   ("knee_db", 0.0, 12.0), ("makeup_gain_db", 0.0, 12.0)],
}


def test_param_tables_match_reference():
    assert list(D.Gain(SR).param_ranges.items()) == [(n, (lo, hi)) for n, lo, hi in EXPECTED["Gain"]]
    assert list(D.Compressor(SR).param_ranges.items()) == [(n, (lo, hi)) for n, lo, hi in EXPECTED["Compressor"]]
    eq = D.ParametricEQ(SR)
    names = list(eq.param_ranges)
    assert eq.num_params == 18 and names[0] == "low_shelf_gain_db" and names[-1] == "high_shelf_q_factor"
    assert names[3:6] == ["band0_gain_db", "band0_cutoff_freq", "band0_q_factor"]
    assert eq.param_ranges["band3_cutoff_freq"] == (12000, 21050) and eq.param_ranges["high_shelf_cutoff_freq"] == (4000, 21050)
    assert eq.param_ranges["low_shelf_cutoff_freq"] == (20, 2000) and eq.param_ranges["band1_cutoff_freq"] == (2000, 8000)
    assert eq.param_ranges["band2_q_factor"] == (0.1, 6.0) and eq.param_ranges["band0_gain_db"] == (-20.0, 20.0)
    rv = D.NoiseShapedReverb(SR)
    assert rv.num_params == 25 and list(rv.param_ranges)[12] == "band0_decay" and list(rv.param_ranges)[-1] == "mix"
    # argument names line up with the functional signatures (process_normalized passes them by keyword)
    import inspect
    for mod in (D.Gain(SR), eq, D.Compressor(SR), D.Expander(SR), rv, D.Distortion()):
        sig = list(inspect.signature(mod.process_fn).parameters)
        assert sig[2:2 + mod.num_params] == list(mod.param_ranges), type(mod).__name__


def test_extract_denormalize_and_errors():
    eq = D.ParametricEQ(SR)
    p = torch.rand(3, 18)
    d = eq.denormalize_param_dict(eq.extract_param_dict(p))
    assert torch.allclose(d["band1_cutoff_freq"], p[:, 7] * 6000 + 2000)
    with pytest.raises(ValueError):
        eq.extract_param_dict(torch.rand(3, 17))
    bad = p.clone(); bad[1, 4] = 1.5
    with pytest.raises(ValueError, match="band0_cutoff_freq"):
        eq.denormalize_param_dict(eq.extract_param_dict(bad))
    with pytest.raises(ValueError, match="band0_cutoff_freq"):
        eq._check_range(bad)
    with pytest.raises(ValueError):
        eq.process_normalized(torch.zeros(3, 1, 8), torch.rand(3, 5))


def test_process_normalized_routes_by_keyword_without_gpu(monkeypatch):
    """The chain of calls, with the kernels stubbed (there is no CPU path to run them)."""
    seen = {}

    def fake(x, sample_rate, **kw):
        seen.update(kw, sample_rate=sample_rate)
        return x
    comp = D.Compressor(SR)
    comp.process_fn = fake
    p = torch.rand(4, 6)
    comp.process_normalized(torch.zeros(4, 2, 16), p)
    assert seen["sample_rate"] == SR and list(seen)[:6] == list(comp.param_ranges)
    assert torch.allclose(seen["ratio"], p[:, 1] * 19 + 1) and torch.allclose(seen["threshold_db"], p[:, 0] * 60 - 60)


def test_range_check_is_one_reduction_and_can_be_delegated():
    """modules.check_unit_range: the reference's message and parameter name (modules.py:83-84), NaN passes as it does there, empty tensors
    pass; inside modules.already_validated() a processor's own check is a no-op (the chain checks all its parameters at once)."""
    from dasp_pytorch_amd import modules as M
    names = ["a", "b", "c"]
    M.check_unit_range(torch.tensor([[0.0, 1.0, 0.5]]), names)
    M.check_unit_range(torch.tensor([[0.5, float("nan"), 0.5]]), names)
    M.check_unit_range(torch.zeros(0, 3), names)
    with pytest.raises(ValueError, match="Parameter c of is out of range."):
        M.check_unit_range(torch.tensor([[0.5, 0.5, 0.5], [0.5, 0.5, -1e-3]]), names)
    with pytest.raises(ValueError, match="Parameter b of is out of range."):
        M.check_unit_range(torch.tensor([[0.5, 1.0001, 2.0]]), names)          # the first offending column is the one named
    g = D.Gain(SR)
    with M.already_validated():
        g._check_range(torch.tensor([[2.0]]))
        with M.already_validated():
            pass
        g._check_range(torch.tensor([[2.0]]))                                  # still inside the outer block
    with pytest.raises(ValueError, match="gain_db"):
        g._check_range(torch.tensor([[2.0]]))
    g.validate_range = False
    g._check_range(torch.tensor([[2.0]]))


def test_stacked_columns_hand_out_contiguous_gradient_rows():
    """functional._StackColumns: torch.stack(cols, 1) whose backward gives each column a contiguous row (autograd then keeps it as .grad
    without a copy); columns that need no gradient get None."""
    from dasp_pytorch_amd.functional import _StackColumns
    cols = [torch.rand(5, 1, dtype=torch.double, requires_grad=(i != 1)) for i in range(3)]
    m = _StackColumns.apply(*cols)
    assert m.shape == (5, 3) and torch.equal(m, torch.stack([c.reshape(-1) for c in cols], 1))
    w = torch.rand(5, 3, dtype=torch.double)
    (m * w).sum().backward()
    assert cols[1].grad is None
    for i in (0, 2):
        assert cols[i].grad.shape == (5, 1) and cols[i].grad.is_contiguous() and torch.equal(cols[i].grad.reshape(-1), w[:, i])
    torch.autograd.gradcheck(lambda *c: _StackColumns.apply(*c), [torch.rand(4, dtype=torch.double, requires_grad=True) for _ in range(3)])


def test_chain_control_tables_follow_the_processors():
    """chain.StyleTransferChain._tables: lo / span of compressor (6), reverb (25), gain (1) in the order dasp_chain_controls reads them,
    rebuilt when a range is edited (the reference reads param_ranges on every call)."""
    from dasp_pytorch_amd.chain import StyleTransferChain
    ch = StyleTransferChain(SR)
    lo, span = ch._tables()
    assert len(lo) == 32 and len(span) == 32
    assert [lo[i] for i in range(6)] == [-60.0, 1.0, 5.0, 5.0, 0.0, 0.0] and [span[i] for i in range(6)] == [60.0, 19.0, 95.0, 95.0, 12.0, 12.0]
    assert all(lo[6 + i] == 0.0 and span[6 + i] == 1.0 for i in range(25)) and (lo[31], span[31]) == (-24.0, 48.0)
    assert ch._tables()[0] is lo                                               # cached
    ch.gain.param_ranges["gain_db"] = (-12.0, 12.0)
    lo2, span2 = ch._tables()
    assert (lo2[31], span2[31]) == (-12.0, 24.0)


def test_reverb_module_passes_its_noise_keywords(monkeypatch):
    """NoiseShapedReverb(device_noise=, noise_seed=, noise_seed_offset=) and StyleTransferChain hand the three to the functional (kernels
    stubbed): what torch.cuda.make_graphed_callables over a chain relies on - a fixed base seed, a device word added to it per replay."""
    from dasp_pytorch_amd import modules as M
    seen = {}

    def fake(x, sample_rate, band_gains, band_decays, mix, num_samples, num_bandpass_taps, noise=None, device_noise=False, noise_seed=None,
             noise_seed_offset=None, decay_bound=0.0):
        seen.update(device_noise=device_noise, noise_seed=noise_seed, noise_seed_offset=noise_seed_offset, decay_bound=decay_bound,
                    num_samples=num_samples, shapes=(tuple(band_gains.shape), tuple(band_decays.shape), tuple(mix.shape)))
        return x
    monkeypatch.setattr(M, "_reverb_from_matrices", fake, raising=False)
    import dasp_pytorch_amd.functional as F
    monkeypatch.setattr(F, "_reverb_from_matrices", fake)
    off = torch.zeros(1, dtype=torch.int64)
    rev = D.NoiseShapedReverb(SR, num_samples=4096, device_noise=True, noise_seed=77, noise_seed_offset=off)
    rev.process_normalized(torch.zeros(3, 2, 64), torch.rand(3, 25))
    assert seen["device_noise"] is True and seen["noise_seed"] == 77 and seen["noise_seed_offset"] is off and seen["num_samples"] == 4096
    assert seen["shapes"] == ((3, 12), (3, 12), (3,))
    assert rev._decay_bound() == pytest.approx(1.0)                          # what the fused (float32, GPU) path vouches for: the validated upper
    rev.validate_range = False                                               # end of the decay range (modules.py:204-230) - nothing once the check is off
    assert rev._decay_bound() == 0.0
    monkeypatch.undo()                                                        # the real function: an offset without generated noise is refused
    with pytest.raises(ValueError, match="noise_seed_offset"):
        F.noise_shaped_reverberation(torch.zeros(1, 2, 8), SR, *[torch.zeros(1)] * 25, noise_seed_offset=off)


def test_decay_bound_is_only_vouched_for_by_a_check_that_ran():
    """NoiseShapedReverb._decay_bound (the promise that lets the filter bank skip a launch) needs a [0, 1] check that has actually looked at
    THIS call's values: the blocking check, or a chain that ran it (already_validated). A deferred check has only been submitted (round 4,
    advisor), a switched-off check says nothing: the bound is then 0 = no promise."""
    import dasp_pytorch_amd as D
    from dasp_pytorch_amd import modules as M
    rev = D.NoiseShapedReverb(44100)
    assert rev._decay_bound() == 1.0
    rev.validate_range = "deferred"
    assert rev._decay_bound() == 0.0
    with M.already_validated():                      # a chain's blocking check covered this call
        assert rev._decay_bound() == 1.0
    with M.already_validated(enforced=False):        # a chain's deferred check was only queued
        assert rev._decay_bound() == 0.0
    rev.validate_range = True
    with M.already_validated(enforced=False):
        assert rev._decay_bound() == 0.0
    assert rev._decay_bound() == 1.0


def test_deferred_range_check_on_cpu_tensors_is_immediate():
    """validate_range = "deferred" exists to avoid a device read-back; CPU parameter tensors are simply checked at once (same ValueError)."""
    import dasp_pytorch_amd as D

    class Probe(D.Processor):
        def __init__(self):
            super().__init__()
            self.sample_rate = 44100
            self.process_fn = lambda x, sr, a, b: x
            self.param_ranges = {"a": (0.0, 1.0), "b": (-1.0, 1.0)}
    m = Probe()
    m.validate_range = "deferred"
    x = torch.zeros(2, 1, 8)
    assert m.process_normalized(x, torch.tensor([[0.1, 0.9], [0.5, 0.5]])) is x
    with pytest.raises(ValueError, match="Parameter b of is out of range"):
        m.process_normalized(x, torch.tensor([[0.1, 0.9], [0.5, 1.5]]))
    m.flush_range_check()                                                   # nothing pending


def test_range_check_is_not_blinded_by_a_nan():
    """The reference's check is (p < 0).any() or (p > 1).any() per parameter (modules.py:83-84): a NaN passes it, an out-of-range value beside
    a NaN does not. A min / max reduction returns NaN for such a column - the check sets NaNs aside first (round 5, advisor)."""
    import torch
    from dasp_pytorch_amd import modules as m
    names = ["a", "b", "c"]
    m.check_unit_range(torch.tensor([[0.5, float("nan"), 1.0], [0.0, 0.3, 0.2]]), names)            # a NaN alone passes
    with pytest.raises(ValueError, match="Parameter b of is out of range"):
        m.check_unit_range(torch.tensor([[0.5, float("nan"), 1.0], [0.0, 1.5, 0.2]]), names)
    with pytest.raises(ValueError, match="Parameter c of is out of range"):
        m.check_unit_range(torch.tensor([[0.5, float("nan"), 1.0], [0.0, float("nan"), -0.1]]), names)
    with pytest.raises(ValueError, match="Parameter c of is out of range"):
        m._DeferredRangeCheck().submit(torch.tensor([[0.5, float("nan"), 2.0]]), names)              # CPU tensors: checked at once
