"""GPU checks of csrc/mtrand.hip: the device reproduction of torch's CPU random stream against torch.randn ITSELF (same
torch.manual_seed, same position in the stream) and against the generator state torch.randn leaves behind (bit-equal).
Tolerance of the values: 1e-6 of the largest sample (the verdict's bar; measured <= 4e-7 absolute: one or two ulps of log, sqrt,
sine and cosine between this arithmetic and the host's vector maths)."""
import numpy as np
import pytest
import torch

from tests.util import record

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def mt():
    assert torch.cuda.is_available()
    from dasp_pytorch_amd import _mt19937
    return _mt19937


def _both(mt, seed, burn, size):
    torch.manual_seed(seed)
    if burn:
        torch.rand(burn)
    s0 = torch.get_rng_state()
    ref = torch.randn(*size)
    s_ref = torch.get_rng_state()
    after_ref = torch.rand(5)
    torch.set_rng_state(s0)
    got = mt.randn_cpu_stream(*size, device=DEV)
    s_got = torch.get_rng_state()
    after_got = torch.rand(5)
    return ref, s_ref, after_ref, got, s_got, after_got


# sizes: one group, the tail rule (numel % 16 != 0) with the tail inside one block / across a block border, draws that stay inside the
# current block, exactly one chunk, several chunks (baby jumps), the reverb's shapes
@pytest.mark.parametrize("burn,size", [
    (0, (16,)), (0, (17,)), (3, (31,)), (0, (624,)), (5, (619,)), (5, (620,)), (100, (16 * 30,)), (609, (40,)), (0, (1000,)), (11, (100003,)),
    (0, (159744,)), (7, (159744 * 2 + 123,)), (0, (2, 12, 5062)), (0, (4, 12, 6000 + 126)), (13, (2, 12, 65536 + 1022)), (1, (16, 12, 66558)),
])
def test_device_stream_equals_torch_randn(mt, burn, size):
    ref, s_ref, after_ref, got, s_got, after_got = _both(mt, 1234, burn, size)
    assert got.shape == ref.shape and got.dtype == torch.float32 and got.device.type == "cuda"
    err = float((got.cpu() - ref).abs().max() / ref.abs().max())
    record(f"mtrand[{burn},{'x'.join(map(str, size))}]", values=err)
    assert err <= 1e-6
    assert torch.equal(s_got, s_ref), "CPU generator state after the device draw differs from the state torch.randn leaves"
    assert torch.equal(after_got, after_ref)


def test_random_positions_and_sizes(mt):
    """40 random (seed, draws already taken, numel) triples - sizes up to five chunks, every alignment of the generator position against
    the 16-element groups and the 624-word blocks left to chance: values and generator state as torch.randn leaves them."""
    rng = np.random.default_rng(2024)
    worst = 0.0
    for _ in range(40):
        seed, burn = int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 3000))
        n = int(rng.choice([rng.integers(16, 2000), rng.integers(2000, 200000), rng.integers(200000, 800000)]))
        ref, s_ref, after_ref, got, s_got, after_got = _both(mt, seed, burn, (n,))
        err = float((got.cpu() - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        assert err <= 1e-6, (seed, burn, n, err)
        assert torch.equal(s_got, s_ref) and torch.equal(after_got, after_ref), (seed, burn, n)
    record("mtrand_random_positions_and_sizes[40]", values=worst)


def test_many_chunks_with_giant_jumps(mt):
    """> 256 chunks (41 M values): the two-phase jump. Compared on slices (the host draw of the whole tensor is the slow thing this
    replaces: ~0.1 s here) and on the generator state."""
    n = 257 * 159744 + 1000 + 5
    ref, s_ref, after_ref, got, s_got, after_got = _both(mt, 99, 17, (n,))
    g = got.cpu()
    for lo in (0, 159744 * 255, 159744 * 256 - 50, 159744 * 256 + 600, n - 4000):
        assert float((g[lo:lo + 4000] - ref[lo:lo + 4000]).abs().max()) <= 4e-6
    assert float((g - ref).abs().max() / ref.abs().max()) <= 1e-6
    assert torch.equal(s_got, s_ref) and torch.equal(after_got, after_ref)


def test_chunks_of_several_units(mt):
    """Beyond 256 units of 256 regenerations (41 M values) a chunk is ceil(units / 256) units, so that no CU gets a second chunk
    (mt_plan): chunk ch starts stride * ch = 256 a + b units on - giant jump a (kept beside the chunks' states), then baby polynomial b.
    514 units: stride 3, 172 chunks; slices around chunk borders, the border next to the giant jump (chunks 85 / 86 start at units 255 /
    258), the whole tensor, and the generator state."""
    n = 513 * 159744 + 77
    ref, s_ref, after_ref, got, s_got, after_got = _both(mt, 3, 5, (n,))
    g = got.cpu()
    for lo in (0, 3 * 159744 - 100, 85 * 3 * 159744 - 300, 86 * 3 * 159744 - 1500, 86 * 3 * 159744 + 17, 256 * 159744 - 50, n - 3000):
        assert float((g[lo:lo + 3000] - ref[lo:lo + 3000]).abs().max()) <= 4e-6
    assert float((g - ref).abs().max() / ref.abs().max()) <= 1e-6
    assert torch.equal(s_got, s_ref) and torch.equal(after_got, after_ref)


def test_one_chunk_per_cu_at_the_largest_reverb_batch(mt):
    """(256, 12, 66558): the noise of BASELINE config 4 - 1,280 units, stride 5, 256 chunks, five giant jumps; chunks that start on a
    giant jump's target (b = 0) are copies."""
    ref, s_ref, after_ref, got, s_got, after_got = _both(mt, 21, 1, (256, 12, 66558))
    g, r = got.cpu().reshape(-1), ref.reshape(-1)
    for lo in (0, 5 * 159744 - 700, 51 * 5 * 159744 - 100, 52 * 5 * 159744 - 100, 256 * 159744 - 100, 1024 * 159744 - 100, g.numel() - 3000):
        assert float((g[lo:lo + 3000] - r[lo:lo + 3000]).abs().max()) <= 4e-6
    assert float((g - r).abs().max() / r.abs().max()) <= 1e-6
    assert torch.equal(s_got, s_ref) and torch.equal(after_got, after_ref)


@pytest.mark.parametrize("n,piece", [(100003, 40000), (65536, 16384), (5000, 4992)])
def test_draws_beyond_one_call_are_made_in_pieces(mt, n, piece):
    """More values than one call of dasp_mt_randn takes (327 M) are drawn in pieces that are multiples of 16, each from the state the last
    one left (read back in between), the tail rule applied to the last piece only and never to fewer than 16 values - forced here with a
    small piece size."""
    torch.manual_seed(11)
    torch.rand(3)
    s0 = torch.get_rng_state()
    ref = torch.randn(n)
    s_ref = torch.get_rng_state()
    words, left = mt.parse_state(s0)
    out = torch.empty(n, device=DEV)
    state_dev, left_after = mt._randn_from_state(words, left, out, max_piece=piece)
    assert float((out.cpu() - ref).abs().max()) <= 4e-6
    assert torch.equal(mt.format_state(s0, state_dev.cpu().numpy().view(np.uint32), left_after), s_ref)


def test_successive_draws_continue_the_stream(mt):
    torch.manual_seed(5)
    a_ref, b_ref, c_ref = torch.randn(1000), torch.rand(7), torch.randn(2, 12, 300)
    torch.manual_seed(5)
    a = mt.randn_cpu_stream(1000, device=DEV)
    b = torch.rand(7)
    c = mt.randn_cpu_stream(2, 12, 300, device=DEV)
    assert float((a.cpu() - a_ref).abs().max()) <= 4e-6 and torch.equal(b, b_ref) and float((c.cpu() - c_ref).abs().max()) <= 4e-6


def test_fallbacks_draw_on_the_host(mt):
    torch.manual_seed(3)
    ref = torch.randn(8)                         # fewer than 16 values: torch takes another code path (double normal_distribution)
    torch.manual_seed(3)
    assert torch.equal(mt.randn_cpu_stream(8, device=DEV).cpu(), ref)
    mt.enabled = False
    try:
        torch.manual_seed(3)
        ref = torch.randn(100)
        torch.manual_seed(3)
        assert torch.equal(mt.randn_cpu_stream(100, device=DEV).cpu(), ref)
    finally:
        mt.enabled = True
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(3)
        ref = torch.randn(100)
        torch.manual_seed(3)
        got = mt.randn_cpu_stream(100, device=DEV)
        assert got.dtype == torch.float64 and torch.equal(got.cpu(), ref)
    finally:
        torch.set_default_dtype(torch.float32)


def test_reverb_default_noise_is_the_reference_stream(mt):
    """functional.noise_shaped_reverberation with default arguments = the same call with the noise torch.randn draws on the host
    (what the reference does, functional.py:548), under the same seed."""
    import dasp_pytorch_amd as D
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(2, 2, 6000, generator=g) * 2 - 1).to(DEV)
    p = [torch.rand(2, generator=g).to(DEV) for _ in range(25)]
    torch.manual_seed(77)
    y_dev = D.noise_shaped_reverberation(x, 44100, *p, num_samples=2048, num_bandpass_taps=127)
    s_dev = torch.get_rng_state()
    torch.manual_seed(77)
    noise = torch.randn(4, 12, 2048 + 126).to(DEV)
    s_host = torch.get_rng_state()
    y_host = D.noise_shaped_reverberation(x, 44100, *p, num_samples=2048, num_bandpass_taps=127, noise=noise)
    assert torch.equal(s_dev, s_host)
    err = float((y_dev - y_host).abs().max() / y_host.abs().max())
    record("reverb_default_noise_device_vs_host", y=err)
    assert err <= 2e-6
