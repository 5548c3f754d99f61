"""TEST INFRASTRUCTURE (oracle): numpy restatement of how torch fills a float32 CPU tensor with normal samples - at::mt19937
(ATen/core/MT19937RNGEngine.h), uniform_real's 24-bit float (ATen/core/TransformationHelper.h) and normal_fill's 16-wide
Box-Muller layout with its tail rule (ATen/native/cpu/DistributionTemplates.h) - which is what the reference's
`torch.randn(bs*2, 12, L + taps - 1)` (dasp_pytorch/functional.py:548) runs. Pinned against torch.randn itself by
tests/test_mt19937_cpu.py; the product (csrc/mtrand.hip) is compared with torch.randn on the GPU box and never imports this."""
import numpy as np

N, M = 624, 397


def _twist(u, v):
    y = (u & np.uint32(0x80000000)) | (v & np.uint32(0x7FFFFFFF))
    return (y >> np.uint32(1)) ^ np.where(v & np.uint32(1), np.uint32(0x9908B0DF), np.uint32(0))


def regenerate(st):
    """mt19937::next_state: one block of 624 words from the previous one."""
    st = np.asarray(st, dtype=np.uint32)
    new = np.empty(N, np.uint32)
    new[0:227] = st[397:624] ^ _twist(st[0:227], st[1:228])
    new[227:454] = new[0:227] ^ _twist(st[227:454], st[228:455])
    new[454:623] = new[227:396] ^ _twist(st[454:623], st[455:624])
    new[623] = new[396] ^ _twist(st[623:624], new[0:1])[0]
    return new


def sequence(st, nblocks):
    """The raw word sequence x[0 .. 624 (nblocks + 1)): the state followed by `nblocks` regenerations."""
    out = [np.asarray(st, dtype=np.uint32)]
    for _ in range(nblocks):
        out.append(regenerate(out[-1]))
    return np.concatenate(out)


def temper(y):
    y = np.asarray(y, dtype=np.uint32).copy()
    y ^= y >> np.uint32(11)
    y ^= (y << np.uint32(7)) & np.uint32(0x9D2C5680)
    y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
    y ^= y >> np.uint32(18)
    return y


def draws(st, left, n):
    """n 32-bit outputs from (state, left) -> (outputs, state afterwards, left afterwards)."""
    rem = left - 1
    cur = np.asarray(st, dtype=np.uint32)
    out = [temper(cur[N - rem:])] if rem else []
    got = rem
    while got < n:
        cur = regenerate(cur)
        out.append(temper(cur))
        got += N
    words = np.concatenate(out) if out else np.zeros(0, np.uint32)
    return words[:n], cur, got - n + 1


def _box_muller16(d):
    d = d.reshape(-1, 16)
    u1 = np.float32(1) - d[:, :8]
    rad = np.sqrt(np.float32(-2) * np.log(u1))
    th = np.float32(2 * np.pi) * d[:, 8:]
    return np.concatenate([rad * np.cos(th), rad * np.sin(th)], 1).astype(np.float32).reshape(-1)


def randn(st, left, n):
    """torch.randn(n) (float32, n >= 16) from (state, left) -> (values, state afterwards, left afterwards)."""
    assert n >= 16
    total = n + (16 if n % 16 else 0)
    w, cur, left_new = draws(st, left, total)
    u = (w & np.uint32(0xFFFFFF)).astype(np.float32) * np.float32(2.0 ** -24)
    data = u[:n].copy()
    k = n // 16
    data[:16 * k] = _box_muller16(data[:16 * k])
    if n % 16:
        data[n - 16:] = _box_muller16(u[n:n + 16])
    return data, cur, left_new
