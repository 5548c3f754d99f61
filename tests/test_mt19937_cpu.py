"""CPU checks of the device reproduction of torch's CPU random stream (dasp_pytorch_amd/_mt19937.py, oracle/mt_stream.py):
the characteristic polynomial and the jump polynomials against plain stepping of the twister, the layout model against
torch.randn and torch's generator state. The kernels themselves are checked on the GPU box (tests/test_gpu_mtrand.py)."""
import numpy as np
import pytest
import torch

from dasp_pytorch_amd import _mt19937 as mt
from oracle import mt_stream as ms


def _exponents(g_row):
    """The set coefficients of a table row (bit s of uint16 k = the coefficient of t^(16 k + s))."""
    bits = np.unpackbits(np.ascontiguousarray(g_row, dtype="<u2").view(np.uint8), bitorder="little")
    return np.nonzero(bits)[0].astype(np.int64)


def _apply(g_row, seq, start):
    """state[w] = XOR over the set exponents i of seq[start + i + w], as the kernel sums it."""
    win = seq[start:start + mt.DEG + mt.N]
    out = np.zeros(mt.N, np.uint32)
    for i in _exponents(g_row):
        out ^= win[i:i + mt.N]
    return out


def test_characteristic_polynomial_annihilates_the_word_sequence():
    p = mt.charpoly()
    exps = [i for i in range(mt.DEG + 1) if (p >> i) & 1]
    assert len(exps) == 135 and exps[-1] == mt.DEG
    st = np.random.default_rng(0).integers(0, 2 ** 32, mt.N, dtype=np.uint64).astype(np.uint32)
    seq = ms.sequence(st, 33)
    for n in (1, 2, 300):
        acc = np.uint32(0)
        for e in exps:
            acc ^= seq[n + e]
        assert acc == 0
    top = 0
    for e in exps:                                  # word 0 of a state only counts with its top bit
        top ^= int(seq[e]) >> 31
    assert top == 0


@pytest.fixture(scope="module")
def table():
    return mt.build_table()


def test_jump_table_layout(table):
    assert table.shape == (mt.N_BABY + mt.N_GIANT, mt.ROW) and table.dtype == np.uint16
    assert mt.GROUP * mt.ROW >= mt.DEG and mt.PAD_INDEX == mt.DEG + mt.N - 1
    for row, g in zip(table[[0, 1, 100, 254, 255, 261]], [mt.jump_polynomials()[i] for i in (0, 1, 100, 254, 255, 261)]):
        ex = _exponents(row)
        assert ex.max() < mt.DEG and 3000 < len(ex) < 11000
        assert sum(1 << int(e) for e in ex) == g


def test_first_jump_polynomials_equal_stepping(table):
    st = np.random.default_rng(1).integers(0, 2 ** 32, mt.N, dtype=np.uint64).astype(np.uint32)
    nb = 2 * mt.BLOCKS_PER_CHUNK
    seq = ms.sequence(st, nb + 34)
    for b in (1, 2):                                # t^J and t^(2J) against 256 and 512 regenerations
        got = _apply(table[b - 1], seq, 0)
        want = seq[b * mt.JUMP:b * mt.JUMP + mt.N]
        assert (got[1:] == want[1:]).all() and (got[0] >> 31) == (want[0] >> 31)


def test_jump_polynomials_compose(table):
    """t^(bJ) applied at an offset equals t^((b+1)J) etc.: baby step 255 followed by step 1 = the first giant step (256 J), and
    giant step a = a times the first one - checked through the windows they select on one sequence (no 40 M-word stepping needed)."""
    F = mt._Field()
    polys = mt.jump_polynomials()
    assert F.mul(polys[254], polys[0]) == polys[255]
    assert F.mul(polys[255], polys[255]) == polys[256]
    assert F.mul(polys[256], polys[255]) == polys[257]
    assert F.mul(polys[260], polys[255]) == polys[261]
    assert F.mul(polys[9], polys[19]) == polys[29]                 # (10 J) + (20 J) = 30 J


@pytest.mark.parametrize("n", [16, 17, 31, 32, 1000, 1008, 624 * 3 + 5, 100003])
@pytest.mark.parametrize("burn", [0, 13, 623, 624])
def test_layout_model_equals_torch_randn(n, burn):
    torch.manual_seed(1234 + n)
    if burn:
        torch.rand(burn)
    s0 = torch.get_rng_state()
    words, left = mt.parse_state(s0)
    ref = torch.randn(n).numpy()
    s1 = torch.get_rng_state()
    got, cur, left_new = ms.randn(words, left, n)
    assert np.abs(got - ref).max() < 1e-6
    total, beta_max, left_after = mt.plan(left, n)
    assert left_after == left_new
    rebuilt = mt.format_state(s0, cur, left_new)
    assert torch.equal(rebuilt, s1)
    assert beta_max == (mt.N - (left - 1) + total - 1) // mt.N


def test_parse_state_rejects_other_layouts():
    with pytest.raises(ValueError):
        mt.parse_state(torch.zeros(100, dtype=torch.uint8))
    with pytest.raises(ValueError):
        mt.parse_state(torch.zeros(mt.STATE_BYTES, dtype=torch.uint8))
