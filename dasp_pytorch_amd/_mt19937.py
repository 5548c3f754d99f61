"""torch's CPU random stream on the device: host side (state parsing, jump-ahead polynomials) of csrc/mtrand.hip.

The reference draws the reverb's white noise with `torch.randn(bs*2, 12, L + taps - 1)` on the GLOBAL CPU GENERATOR whatever
the device of `x` (dasp_pytorch/functional.py:548) - at (128,2,262144) that is 204 M values drawn by one host thread (0.56 s)
and 0.8 GB copied to the device, against 2.3 ms of kernels. `torch.manual_seed` parity with the reference needs exactly that
stream, so the stream is reproduced where it is needed:

* the generator is the 32-bit Mersenne Twister (ATen/core/MT19937RNGEngine.h): state 624 words, `left` words of the current
  block not yet handed out (+ 1), regenerated 624 words at a time;
* a float draw keeps the low 24 bits of one 32-bit output: u = (y & 0xFFFFFF) * 2^-24 (ATen/core/TransformationHelper.h
  `uniform_real`);
* `normal_` on a contiguous float tensor of >= 16 elements (ATen/native/cpu/DistributionTemplates.h `normal_fill`) first fills
  every element with one such draw, then turns every aligned group of 16 into 8 Box-Muller pairs (element j with element j + 8:
  radius from 1 - u[j], angle from u[j + 8], cosine to j, sine to j + 8) and, when numel % 16 != 0, recomputes the LAST 16
  elements from 16 fresh draws.

The twister is linear over GF(2): the state after J more words is g_J(T) applied to the state, g_J = t^J mod p(t), p the
characteristic polynomial (degree 19937, 135 terms), and because every bit of the word sequence x[n] is a linear functional of
the state, x[n + J] = XOR over the set coefficients i of g_J of x[n + i]. That turns "jump ahead" into sums over a window of the
sequence itself, which a workgroup holds in LDS (csrc/mtrand.hip, mt_jump_kernel); with the states of all chunks known, the
chunks regenerate their blocks side by side. This module computes p (closed form of Matsumoto & Nishimura 1998, app. A) and the
table of g_J with plain Python integers as GF(2)[t] polynomials, caches it next to the library, and moves generator states
between torch's byte layout and the kernels.

tests/test_mt19937_cpu.py checks p against the recurrence, the jump polynomials against stepping, and the numpy model of the
layout (oracle/mt_stream.py) against torch.randn; tests/test_gpu_mtrand.py checks the kernels against torch.randn itself."""
import os
import struct

import numpy as np

N, M = 624, 397
DEG = 19937
MATRIX_A = 0x9908B0DF

# layout constants shared with csrc/mtrand.hip (dasp_mt_layout() reports the kernel's side; the binding compares)
BLOCKS_PER_CHUNK = 256                  # a unit = this many regenerations = 159,744 words; a chunk (one workgroup of the generation kernel) = 1 .. 8 units
JUMP = N * BLOCKS_PER_CHUNK             # words per unit: the jump polynomials are powers of t^JUMP
N_BABY = 255                            # t^(b J), b = 1 .. 255: the state 256 a + b units on from the state 256 a units on
N_GIANT = 7                             # t^(a 256 J), a = 1 .. 7: the state 256 a units on from the start state  => at most 2048 units (327 M draws) per call
GROUP = 16                              # exponents per group: bit s of uint16 g of a row = coefficient of t^(16 g + s)
ROW = 1280                              # uint16 per polynomial: 1,247 groups, padded with zeros
PAD_INDEX = 20560                       # first word behind the sequence window of a jump (19937 + 623): the kernel keeps zeros there
MAX_CHUNKS = (N_GIANT + 1) * (N_BABY + 1)

_HERE = os.path.dirname(os.path.abspath(__file__))
TABLE_PATH = os.path.join(_HERE, "csrc", "mt_jump_table.npy")        # built artefact (git-ignored, travels to the GPU box like the .so)


# ---- GF(2)[t] with Python integers (bit i = coefficient of t^i) ----------------------------------------------------------------
def _clmul(x, y):
    if bin(y).count("1") > bin(x).count("1"):
        x, y = y, x
    acc = 0
    while y:
        low = y & -y
        acc ^= x << (low.bit_length() - 1)
        y ^= low
    return acc


def charpoly():
    """p(t) of the MT19937 state transition (Matsumoto & Nishimura 1998, appendix A.1), n = 624, m = 397, w = 32, r = 31, a = MATRIX_A:
    (t^n + t^m)^(w-r) (t^(n-1) + t^(m-1))^r + sum_{j<r} a_j (t^n + t^m)^(w-r) (t^(n-1) + t^(m-1))^(r-j-1) + sum_{j>=r} a_j (t^n + t^m)^(w-j-1)."""
    w, r = 32, 31
    A = (1 << N) | (1 << M)
    Bp = (1 << (N - 1)) | (1 << (M - 1))
    pw = [1]
    for _ in range(r):
        pw.append(_clmul(pw[-1], Bp))          # pw[k] = (t^(n-1) + t^(m-1))^k
    p = _clmul(A, pw[r])                       # w - r = 1
    for j in range(r):
        if (MATRIX_A >> j) & 1:
            p ^= _clmul(A, pw[r - j - 1])
    for j in range(r, w):
        if (MATRIX_A >> j) & 1:
            p ^= 1                             # (t^n + t^m)^0, j = w - 1 only
    assert p.bit_length() - 1 == DEG
    return p


class _Field:
    def __init__(self):
        self.p = charpoly()
        self.low = [i for i in range(DEG) if (self.p >> i) & 1]       # p = t^19937 + sum t^e
        self.mask = (1 << DEG) - 1

    def reduce(self, a):
        while a.bit_length() > DEG:
            hi = a >> DEG
            a &= self.mask
            for e in self.low:
                a ^= hi << e
        return a

    def mul(self, a, b):
        return self.reduce(_clmul(a, b))

    def tpow(self, e):
        res, base = 1, 2
        while e:
            if e & 1:
                res = self.mul(res, base)
            e >>= 1
            if e:
                base = self.mul(base, base)
        return res


def jump_polynomials():
    """[g for the 255 baby steps] + [g for the 7 giant steps], as Python integers."""
    F = _Field()
    g1 = F.tpow(JUMP)
    baby = [g1]
    for _ in range(N_BABY - 1):
        baby.append(F.mul(baby[-1], g1))
    G1 = F.mul(baby[-1], g1)                   # t^(256 J)
    giant = [G1]
    for _ in range(N_GIANT - 1):
        giant.append(F.mul(giant[-1], G1))
    return baby + giant


def _expand(g):
    """One polynomial as the kernel reads it: ROW uint16, bit s of word k = the coefficient of t^(16 k + s)."""
    row = np.zeros(ROW, dtype=np.uint16)
    raw = np.frombuffer(g.to_bytes(2 * ((DEG + 15) // 16), "little"), dtype="<u2")
    row[:len(raw)] = raw
    return row


def build_table(path=TABLE_PATH, force=False):
    """(N_BABY + N_GIANT, ROW) uint16, cached at `path` (a few seconds of integer arithmetic the first time)."""
    if not force and os.path.exists(path):
        try:
            t = np.load(path)
            if t.shape == (N_BABY + N_GIANT, ROW) and t.dtype == np.uint16:
                return t
        except Exception:
            pass
    t = np.stack([_expand(g) for g in jump_polynomials()])
    try:
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            np.save(f, t)
        os.replace(tmp, path)
    except OSError:
        pass                                    # read-only tree: keep it in memory
    return t


_table_host = None
_table_dev = {}


def table(device):
    """The expanded table on `device` (uploaded once per device and process)."""
    import torch
    global _table_host
    key = torch.device(device)
    t = _table_dev.get(key)
    if t is None:
        if _table_host is None:
            _table_host = build_table()
        t = torch.from_numpy(_table_host.view(np.int16)).to(key)
        _table_dev[key] = t
    return t


# ---- torch's CPU generator state (at::CPUGeneratorImplState, legacy layout first) -------------------------------------------------
# uint64 seed | int32 left | int32 seeded | uint64 next | uint64 state[624] | double normal_x, normal_y, normal_rho | int32 normal_is_valid |
# pad | float next_float_normal_sample | bool valid | pad   = 5056 bytes
_OFF_LEFT, _OFF_NEXT, _OFF_STATE, STATE_BYTES = 8, 16, 24, 5056


def parse_state(state):
    """torch.get_rng_state() -> (words uint32[624], left). Raises on a layout this module does not know."""
    b = state.numpy().tobytes()
    if len(b) != STATE_BYTES:
        raise ValueError(f"unexpected CPU generator state of {len(b)} bytes (known layout: {STATE_BYTES})")
    left, seeded = struct.unpack_from("<ii", b, _OFF_LEFT)
    nxt, = struct.unpack_from("<q", b, _OFF_NEXT)
    words = np.frombuffer(b, dtype=np.uint64, count=N, offset=_OFF_STATE)
    if not (1 <= left <= N and seeded == 1 and (nxt + left == N + 1 or (left == 1 and nxt == 0)) and int(words.max()) < (1 << 32)):
        raise ValueError("CPU generator state does not look like a seeded at::mt19937")
    return words.astype(np.uint32), left


def format_state(state, words, left):
    """A copy of `state` (torch.get_rng_state()) with the twister's words and position replaced; everything else (seed, the cached
    double / float normal samples) is kept as it was - normal_fill does not touch those either."""
    import torch
    b = bytearray(state.numpy().tobytes())
    struct.pack_into("<i", b, _OFF_LEFT, int(left))
    struct.pack_into("<q", b, _OFF_NEXT, N + 1 - int(left))
    b[_OFF_STATE:_OFF_STATE + 8 * N] = np.asarray(words, dtype=np.uint32).astype(np.uint64).tobytes()
    return torch.frombuffer(b, dtype=torch.uint8).clone()


def plan(left, n):
    """Draw bookkeeping for `n` normal values from a generator with `left`: (total draws, last block, left afterwards).
    Word w of the sequence (w = 0 .. 623: the current state) is draw w - (624 - rem), rem = left - 1."""
    rem = left - 1
    total = n + (16 if n % 16 else 0)
    last_word = N - rem + total - 1
    beta_max = last_word // N                   # regenerations needed
    left_after = N * (beta_max + 1) - last_word
    return total, beta_max, left_after


# ---- the draw ---------------------------------------------------------------------------------------------------------------------
enabled = True            # False: every draw on the host, as before round 6 (bench.py's A/B and the tests of the fallback set it)
_layout_checked = False
_self_check = {}          # device -> True (the kernels reproduced torch.randn and the generator state on this device) / False (they did not)


def _check_layout():
    global _layout_checked
    if not _layout_checked:
        import ctypes
        from . import _lib
        out = (ctypes.c_int * 8)()
        _lib.call("dasp_mt_layout", out)
        want = [BLOCKS_PER_CHUNK, N_BABY, N_GIANT, GROUP, ROW, PAD_INDEX, MAX_CHUNKS, 0]
        if list(out) != want:
            raise _lib.DaspHipError(f"jump-table layout of libdasp_hip.so {list(out)} != _mt19937.py {want}")
        _layout_checked = True


def _randn_from_state(words, left, out, max_piece=None):
    """Fill the flat float32 device tensor `out` from (words, left); returns (words afterwards as a device tensor or None when the draw
    stayed inside the current block, left afterwards). Asynchronous on the current stream - except beyond one call's capacity (327 M
    values; `max_piece`: a smaller bound, for the tests): the draw is then made in pieces, each starting from the state the previous one
    left, which is read back in between."""
    import ctypes

    import torch

    from . import _lib
    _check_layout()
    n = out.numel()
    max_n = min(int(_lib.lib().dasp_mt_max_values()), int(max_piece) if max_piece else 1 << 62) // 16 * 16
    tab = table(out.device)
    state_dev, done = None, 0
    host_words = np.ascontiguousarray(words, dtype=np.uint32)
    while done < n:
        piece = min(n - done, max_n)
        if 0 < n - done - piece < 16:            # never leave a last piece below the 16 values normal_fill's layout needs
            piece -= 16
        if state_dev is not None:                # (pieces beyond 327 M values: the next piece starts from the state the last one left - one read-back)
            host_words = state_dev.cpu().numpy().view(np.uint32)
        words_needed = _lib.lib().dasp_mt_scratch_words(int(left), ctypes.c_longlong(piece))
        scratch = torch.empty(words_needed, dtype=torch.int32, device=out.device)
        left_after, regen, off = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_long(0)
        _lib.call("dasp_mt_randn", host_words.ctypes.data_as(ctypes.c_void_p), int(left), ctypes.c_void_p(out.data_ptr() + 4 * done),
                  ctypes.c_longlong(piece), _lib.ptr(tab), _lib.ptr(scratch), ctypes.byref(left_after), ctypes.byref(regen), ctypes.byref(off), _lib.stream())
        if regen.value:
            state_dev = scratch[off.value:off.value + N]
        left = left_after.value
        done += piece
    return state_dev, left


class _PendingState:
    """The generator state a device draw leaves behind, on its way to the host: the words were copied to pinned memory right behind the
    generating kernels (an event marks the copy), `finish()` waits for that event only and installs the state. Between the draw and
    finish() the CPU generator still holds the state from BEFORE the draw - the caller finishes before it hands control back."""

    def __init__(self, state, words, left_after, state_dev):
        import torch
        self.state, self.words, self.left_after, self.pinned, self.event = state, words, left_after, None, None
        if state_dev is not None:
            self.pinned = torch.empty(N, dtype=torch.int32, pin_memory=True)
            self.pinned.copy_(state_dev, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()

    def finish(self):
        import torch
        if self.state is None:
            return
        if self.event is not None:
            self.event.synchronize()
            self.words = self.pinned.numpy().view(np.uint32)
        torch.set_rng_state(format_state(self.state, self.words, self.left_after))
        self.state = None


class _NothingPending:
    def finish(self):
        pass


def randn_cpu_stream(*size, device, defer=False):
    """`torch.randn(*size)` of the reference's call site (dasp_pytorch/functional.py:548: float32, default CPU generator) delivered on
    `device`: the values the CPU call would have returned, and the CPU generator left in the state that call would have left it in -
    generated by csrc/mtrand.hip from the generator's state instead of drawn on the host and copied. Falls back to the host draw where
    the reproduction does not apply (fewer than 16 values, a default dtype other than float32, stream capture - the draw needs one
    read-back of 2.5 KB for the state -, a generator state of an unknown layout, or a failed self-check on this device).
    defer=True returns (tensor, pending): the state's read-back is on its way and `pending.finish()` installs it - for a caller that
    queues more device work first (the reverb's kernels) so that the host does not sit out the generation; it must finish before
    anything else can draw from the CPU generator. (torch.randn holds the generator's mutex for the whole fill; between this function's
    get_rng_state and set_rng_state another host THREAD drawing from the global generator would be overwritten - as racy as two threads
    sharing one seeded stream are in the reference, but with a wider window.)"""
    import torch
    dev = torch.device(device)
    n = 1
    for s in size:
        n *= int(s)

    def host():
        t = torch.randn(*size).to(dev)
        return (t, _NothingPending()) if defer else t

    if not enabled or n < 16 or torch.get_default_dtype() != torch.float32 or dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
        return host()
    if _self_check.get(dev) is None:
        from . import _lib
        with torch.cuda.device(dev):
            _lib.check(-3 if _lib.lib().dasp_device_error() else 0, "dasp_mt_randn")       # (a pending device error is not a failed self-check)
        _self_check[dev] = _run_self_check(dev)
        if not _self_check[dev]:
            import warnings
            warnings.warn("dasp_pytorch_amd: the device reproduction of torch's CPU random stream does not match torch.randn of this torch "
                          f"build ({torch.__version__}); the reverb's default noise is drawn on the host instead (slow). "
                          "noise_shaped_reverberation(..., device_noise=True) is unaffected.")
    if not _self_check[dev]:
        return host()
    state = torch.get_rng_state()
    try:
        words, left = parse_state(state)
    except ValueError:
        return host()
    with torch.cuda.device(dev):
        out = torch.empty(n, dtype=torch.float32, device=dev)
        state_dev, left_after = _randn_from_state(words, left, out)
        pending = _PendingState(state, words, left_after, state_dev)
    if defer:
        return out.view(*size), pending
    pending.finish()
    return out.view(*size)


def _run_self_check(dev):
    """Once per device and process: 16 K + 3 values (several chunks' worth would cost a second; this covers regeneration, groups across
    block borders and the tail rule) from a scratch copy of the generator against torch.randn itself, and the state afterwards."""
    import torch
    keep = torch.get_rng_state()
    try:
        ok = True
        for seed, burn, n in ((20240601, 0, 16387), (7, 100, 4096)):
            torch.manual_seed(seed)
            if burn:
                torch.rand(burn)
            s0 = torch.get_rng_state()
            ref = torch.randn(n)
            s1 = torch.get_rng_state()
            words, left = parse_state(s0)
            with torch.cuda.device(dev):
                out = torch.empty(n, dtype=torch.float32, device=dev)
                state_dev, left_after = _randn_from_state(words, left, out)
                new_words = state_dev.cpu().numpy().view(np.uint32) if state_dev is not None else words
            ok = ok and bool((out.cpu() - ref).abs().max() <= 2e-6) and torch.equal(format_state(s0, new_words, left_after), s1)
        return ok
    except Exception:
        return False
    finally:
        torch.set_rng_state(keep)
