"""Test infrastructure (not imported by the product): numpy model of the counter-based white-noise generator of csrc/reverb.hip
(`device_noise=True`: the noise of noise_shaped_reverberation, dasp_pytorch/functional.py:548, generated inside the filter-bank kernels
instead of being written to memory by torch.randn and read back twice).

Stream definition. For signal pair b (batch item), band k and sample index m, ONE 32-bit hash gives the two rows of the item (the
reference's noise rows 2b and 2b+1, i.e. left / right) as a Box-Muller pair:
    key    = mix(seed, b * nb + k)         (a, c) = (24-bit odd multiplier, 32-bit offset)
    h      = lowbias32(m * a + c)          (Wellons' 2-round multiply-xorshift finaliser; m < 2^24)
    u1     = ((h >> 16) + 0.5) / 65536,  u2 = (h & 0xffff) / 65536
    r      = sqrt(-2 ln u1);   noise[2b, k, m] = r cos(2 pi u2),  noise[2b+1, k, m] = r sin(2 pi u2)
(the 16-bit radius grid has E r^2 / 2 = 0.999995.) Distinct (b, k) get distinct affine index maps into the hash, so two streams
share at most a handful of isolated values, never a run."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def lowbias32(x):
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7feb352d)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846ca68b)) & M32
    x ^= x >> np.uint64(16)
    return x


def stream_key(seed, sid):
    """(a, c) of stream `sid` = b * nb + band under the 64-bit seed."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    k0 = lowbias32(np.uint64(sid) + np.uint64(0x9E3779B9))
    k1 = lowbias32(k0 ^ lo)
    k2 = lowbias32(k1 ^ hi ^ np.uint64(0x85EBCA6B))
    a = (k1 & np.uint64(0xFFFFFF)) | np.uint64(1)
    return a, k2


def noise_pair(seed, sid, m):
    a, c = stream_key(seed, sid)
    h = lowbias32((np.asarray(m, dtype=np.uint64) * a + c) & M32)
    u1 = ((h >> np.uint64(16)).astype(np.float64) + 0.5) / 65536.0
    u2 = (h & np.uint64(0xFFFF)).astype(np.float64) / 65536.0
    r = np.sqrt(-2.0 * np.log(u1))
    return r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)


def noise(seed, B, nb, row_len):
    """(2B, nb, row_len) float64, the layout of the reference's torch.randn(bs * 2, 12, num_samples + taps - 1)."""
    out = np.empty((2 * B, nb, row_len))
    m = np.arange(row_len)
    for b in range(B):
        for k in range(nb):
            out[2 * b, k], out[2 * b + 1, k] = noise_pair(seed, b * nb + k, m)
    return out
