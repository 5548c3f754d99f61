"""fp64 numpy model of the chunked-scan biquad-cascade algorithm the HIP kernels implement.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing in the product package imports this.
It is *not* a restatement of the reference (that is oracle/dasp_oracle.py); it is an executable
specification of csrc/sosfilt.hip -- same realisation, same tables, same scans, same gradient
correlations -- used by the CPU tests to prove the algorithm equals the reference's
`sosfilt_via_fsm` (dasp_pytorch/signal.py:136-166) and its autograd.

Realisation of one section  H(z) = (b0 + b1 z^-1 + b2 z^-2) / (1 + a1 z^-1 + a2 z^-2):
    s1' = sg*s1 - kom*s2 + u        (A = [[sg, -kom], [om, sg]], B = [1, 0])
    s2' = om*s1 + sg*s2
    y   = g1*s1 + g2*s2 + d*u       (C = [g1, g2], D = d = b0)
with sg = -a1/2, disc = sg^2 - a2, om = max(sqrt|disc|, OM_MIN), kom = sign(-disc)*om
(rotation-scaling matrix for complex poles, symmetric matrix for real poles: normal either way),
g1 = b1 - b0*a1, g2 = ((b2 - b0*a2) + g1*sg)/om.  Then s2[n] = om * w[n-2] where w = u / A(z).
The lane scans (forward and adjoint) use this realisation; the adjoint cascade inside a chunk runs each section in transposed
direct form II (backward_row). The forward cascade inside a chunk uses it for sections whose poles are real or closer than
DF_OM_MIN (imaginary part) to the real axis; the others ("direct" sections) can run in a direct form from the chunk's exact start
state (s1, s2), mapped once per chunk:
    forward kernel (only when built with DASP_FWD_DIRECT; FWD_DIRECT here), transposed form II:
        z1 = g1 s1 + g2 s2,  z2 = zc1 s1 + zc2 s2  with (zc1, zc2) = C (A + a1 I)
    backward kernel's recomputation, form II:  w[-2] = s2 / om,  w[-1] = s1 + (sg / om) s2; the kept signal is w itself, so the
    coefficient correlations of a direct section are not divided by om.

Coefficient correlations (backward_row, csrc/sosfilt.hip `adjoint` / finalize_section). With K[n] the kept signal of a section (w[n - 2]
for a direct section, om w[n - 2] for a normal-form one), g its adjoint input and o its adjoint output, the kernel sums
    direct section:       sum g K[n+2],  sum g K[n+1],  sum g K[n],  sum o K[n+1],  sum o K[n]
    normal-form section:  sum g D[n+1],  sum g D[n],    sum g K[n],  sum o D[n],    sum o K[n]      with D[n] = K[n+1] - sg K[n]
(the lags of K agree to many digits when the poles sit near z = 1; D is their small difference, formed before the summation instead of
after it) and the finalize step maps them back. fast=True is the kernel variant for cascades that come from the RBJ design (b0 > 0):
every section is recomputed in monic form (feed-through 1, its signals scaled by 1 / product of the earlier b0), the last section
computes no output, the lag-0 sum of every section is left out and recovered from T = <adjoint output, input> of the last section,
because sum_i b_i (sum g w[n - i]) = <g, y> is the same number for every section of a cascade.
"""
import numpy as np

OM_MIN = 1e-5
DF_OM_MIN = 0.125      # csrc/sosfilt.hip DASP_DF_OM_MIN
FWD_DIRECT = False     # csrc/sosfilt.hip DASP_FWD_DIRECT: the forward kernel keeps every section in normal form by default
WAVE = 64


def realize(sos):
    """sos (S,6) -> dict of per-section arrays (a0-normalised)."""
    sos = np.asarray(sos, np.float64)
    a0 = sos[:, 3]
    b0, b1, b2 = sos[:, 0] / a0, sos[:, 1] / a0, sos[:, 2] / a0
    a1, a2 = sos[:, 4] / a0, sos[:, 5] / a0
    sg = -a1 / 2
    disc = sg * sg - a2
    om = np.maximum(np.sqrt(np.abs(disc)), OM_MIN)
    kom = np.where(disc < 0, om, -om)
    g1 = b1 - b0 * a1
    g2 = ((b2 - b0 * a2) + g1 * sg) / om
    direct = (disc < 0) & (om >= DF_OM_MIN) & ~((b0 == 1.0) & (b1 == a1) & (b2 == a2))   # identity sections stay in normal form (exact)
    return dict(sg=sg, om=om, kom=kom, g1=g1, g2=g2, d=b0, b=np.stack([b0, b1, b2], 1), a=np.stack([a1, a2], 1), direct=direct,
                zc1=-g1 * sg + g2 * om, zc2=-g1 * kom - g2 * sg)


def _sections_fwd(r):
    S = len(r["sg"])
    return [(np.array([[r["sg"][k], -r["kom"][k]], [r["om"][k], r["sg"][k]]]), np.array([1.0, 0.0]),
             np.array([r["g1"][k], r["g2"][k]]), r["d"][k]) for k in range(S)]


def _sections_adj(r):
    """Adjoint cascade: sections in reverse order, (A^T, C^T, B^T, d), runs on reversed time."""
    out = []
    for (A, B, C, d) in reversed(_sections_fwd(r)):
        out.append((A.T.copy(), C.copy(), B.copy(), d))
    return out


def cascade_system(secs):
    """Block lower-triangular transition matrix Phi (2S,2S) and input vector Bx (2S)."""
    S = len(secs)
    Phi = np.zeros((2 * S, 2 * S))
    Bx = np.zeros(2 * S)
    for k, (A, B, C, d) in enumerate(secs):
        Phi[2 * k:2 * k + 2, 2 * k:2 * k + 2] = A
        gain = 1.0  # product of d_i for j < i < k
        for j in range(k - 1, -1, -1):
            Cj = secs[j][2]
            Phi[2 * k:2 * k + 2, 2 * j:2 * j + 2] = np.outer(B, Cj) * gain
            gain *= secs[j][3]
        Bx[2 * k:2 * k + 2] = B * gain
    return Phi, Bx


def chunk_tables(secs, L):
    """G (2S,L): chunk zero-state end state = G @ x_chunk.  M = Phi^L.  P[k][l] = (M_kk)^(2^l)."""
    S = len(secs)
    Phi, Bx = cascade_system(secs)
    G = np.zeros((2 * S, L))
    v = Bx.copy()
    for n in range(L - 1, -1, -1):
        G[:, n] = v
        v = Phi @ v
    M = np.linalg.matrix_power(Phi, L)
    P = np.zeros((S, 6, 2, 2))
    for k in range(S):
        Q = M[2 * k:2 * k + 2, 2 * k:2 * k + 2].copy()
        for l in range(6):
            P[k, l] = Q
            Q = Q @ Q
    return G, M, P


def tile_scan(z, M, P, carry):
    """z (64,2S) chunk zero-state end states; carry (2S,) state at tile start.
    Returns (start (64,2S) state at each chunk start, carry_out (2S,))."""
    S = z.shape[1] // 2
    start = np.zeros_like(z)
    out = np.zeros(2 * S)
    lane = np.arange(WAVE)
    for k in range(S):
        f = z[:, 2 * k:2 * k + 2].copy()
        for j in range(k):
            f += start[:, 2 * j:2 * j + 2] @ M[2 * k:2 * k + 2, 2 * j:2 * j + 2].T
        f[0] += P[k, 0] @ carry[2 * k:2 * k + 2]
        E = f
        for l in range(6):
            sh = 1 << l
            t = np.zeros_like(E)
            t[sh:] = E[:-sh]
            E = E + np.where((lane >= sh)[:, None], t @ P[k, l].T, 0.0)
        out[2 * k:2 * k + 2] = E[WAVE - 1]
        start[1:, 2 * k:2 * k + 2] = E[:-1]
        start[0, 2 * k:2 * k + 2] = carry[2 * k:2 * k + 2]
    return start, out


def cascade_chunks(secs, X, start, store_s2=False, r=None, monic=False):
    """Run the cascade over each lane's chunk X (64,L) from start states (64,2S), one section at a time over the chunk as the
    kernels do. Returns Y (64,L) and (optionally) S2 (S,64,L+2): for a normal-form section s2_k[n] for n in chunk plus the two
    states following the chunk (s2 does not depend on the current input); for a direct section (r["direct"], when r is given)
    w_k[n-2] instead (= s2_k[n] / om)."""
    S = len(secs)
    L = X.shape[1]
    U = X.copy()
    S2 = np.zeros((S, WAVE, L + 2)) if store_s2 else None
    if monic:        # the backward kernel's FAST recomputation: returns (input of the last section / P_{S-1}, None, kept signals / P_k)
        assert store_s2 and r is not None
        q = 1.0
        for k, (A, B, C, d) in enumerate(secs):
            s1, s2 = start[:, 2 * k] * q, start[:, 2 * k + 1] * q
            last = k == S - 1
            if r["direct"][k]:
                b0, b1, b2 = r["b"][k]
                a1, a2 = r["a"][k]
                w2 = s2 / r["om"][k]
                w1 = s1 + (r["sg"][k] / r["om"][k]) * s2
                for n in range(L):
                    u = U[:, n].copy()
                    S2[k, :, n] = w2
                    w = u - a1 * w1 - a2 * w2
                    if not last:
                        U[:, n] = w + (b1 / b0) * w1 + (b2 / b0) * w2
                    w2, w1 = w1, w
                S2[k, :, L] = w2
            else:
                for n in range(L):
                    u = U[:, n].copy()
                    S2[k, :, n] = s2
                    if not last:
                        U[:, n] = (C[0] / d) * s1 + (C[1] / d) * s2 + u
                    s1, s2 = A[0, 0] * s1 + A[0, 1] * s2 + u, A[1, 0] * s1 + A[1, 1] * s2
                S2[k, :, L] = s2
            q = q / d
        return U, None, S2
    for k, (A, B, C, d) in enumerate(secs):
        s1, s2 = start[:, 2 * k].copy(), start[:, 2 * k + 1].copy()
        direct = r is not None and bool(r["direct"][k])
        if direct and not store_s2:      # forward kernel: transposed direct form II
            b0, b1, b2 = r["b"][k]
            a1, a2 = r["a"][k]
            z1 = r["g1"][k] * s1 + r["g2"][k] * s2
            z2 = r["zc1"][k] * s1 + r["zc2"][k] * s2
            for n in range(L):
                u = U[:, n]
                o = b0 * u + z1
                z1 = b1 * u - a1 * o + z2
                z2 = b2 * u - a2 * o
                U[:, n] = o
        elif direct:                     # backward kernel: direct form II, keeping w
            b0, b1, b2 = r["b"][k]
            a1, a2 = r["a"][k]
            w2 = s2 / r["om"][k]
            w1 = s1 + (r["sg"][k] / r["om"][k]) * s2
            for n in range(L):
                u = U[:, n]
                S2[k, :, n] = w2
                w = u - a1 * w1 - a2 * w2
                U[:, n] = b0 * w + b1 * w1 + b2 * w2
                w2, w1 = w1, w
            S2[k, :, L] = w2
            S2[k, :, L + 1] = w1
        else:
            for n in range(L):
                u = U[:, n].copy()
                if store_s2:
                    S2[k, :, n] = s2
                U[:, n] = C[0] * s1 + C[1] * s2 + d * u
                s1, s2 = A[0, 0] * s1 + A[0, 1] * s2 + B[0] * u, A[1, 0] * s1 + A[1, 1] * s2 + B[1] * u
            if store_s2:
                S2[k, :, L] = s2
                S2[k, :, L + 1] = A[1, 0] * s1 + A[1, 1] * s2
    return U, None, S2


def forward_row(r, x, L, save_every=None, tiles=None, carry0=None, scan_only=False):
    """Forward filter one row x (N,) with realisation r. Returns y and saved tile carries.
    tiles = (t0, t1) restricts the pass to that tile range starting from the cascade state carry0 (the segmented scheme below);
    scan_only skips the cascade (no output) and returns the state after tile t1 - 1 instead of y."""
    secs = _sections_fwd(r)
    G, M, P = chunk_tables(secs, L)
    S = len(secs)
    TS = WAVE * L
    N = len(x)
    nt = (N + TS - 1) // TS
    xp = np.zeros(nt * TS)
    xp[:N] = x
    y = np.zeros(nt * TS)
    t0, t1 = tiles if tiles is not None else (0, nt)
    carry = np.zeros(2 * S) if carry0 is None else np.asarray(carry0, np.float64).copy()
    carries = np.zeros((nt, 2 * S))
    for t in range(t0, t1):
        X = xp[t * TS:(t + 1) * TS].reshape(WAVE, L)
        carries[t] = carry
        z = X @ G.T
        start, carry = tile_scan(z, M, P, carry)
        if scan_only:
            continue
        Y, _, _ = cascade_chunks(secs, X, start, r=r if FWD_DIRECT else None)
        y[t * TS:(t + 1) * TS] = Y.reshape(-1)
    if scan_only:
        return carry, carries
    return (y if tiles is not None else y[:N]), carries


def backward_row(r, x, gy, carries, L, tiles=None, acarry0=None, scan_only=False, raw=False, fast=False):
    """Backward for one row: returns gx (N,), and gb (S,3), ga (S,3) -- gradients w.r.t. the
    a0-normalised coefficients (ga[:,0] is d/da0 from scale invariance).
    tiles = (t0, t1) restricts the pass to tiles t1 - 1 .. t0 starting from the adjoint cascade state acarry0; scan_only runs only the
    adjoint lane scans and returns the adjoint state below tile t0; raw returns (gx padded, acc_b, acc_a) un-normalised so that
    segments can be summed (the segmented scheme below)."""
    fs = _sections_fwd(r)
    ads = _sections_adj(r)
    S = len(fs)
    G, M, P = chunk_tables(fs, L)
    Ga, Ma, Pa = chunk_tables(ads, L)
    TS = WAVE * L
    N = len(x)
    nt = (N + TS - 1) // TS
    xp = np.zeros(nt * TS); xp[:N] = x
    gp = np.zeros(nt * TS); gp[:N] = gy
    gx = np.zeros(nt * TS)
    acc = np.zeros((S, 5))     # the kernel's five sums per section (module docstring); fast: [:, 0] = T
    t0, t1 = tiles if tiles is not None else (0, nt)
    acarry = np.zeros(2 * S) if acarry0 is None else np.asarray(acarry0, np.float64).copy()
    for t in range(t1 - 1, t0 - 1, -1):
        X = xp[t * TS:(t + 1) * TS].reshape(WAVE, L)
        GY = gp[t * TS:(t + 1) * TS].reshape(WAVE, L)
        # adjoint: same machinery on (lane, sample)-reversed data
        GYr = GY[::-1, ::-1]
        astart_r, acarry = tile_scan(GYr @ Ga.T, Ma, Pa, acarry)
        if scan_only:
            continue
        # forward chunk start states from the saved tile carry
        start, _ = tile_scan(X @ G.T, M, P, carries[t])
        Ulast, _, S2 = cascade_chunks(fs, X, start, store_s2=True, r=r, monic=fast)
        # per-lane adjoint cascade with correlations (natural lane order, descending n). As in the kernel, each adjoint section
        # runs in transposed direct form II from the chunk's entry costate (normal-form coordinates, from the scan), mapped once
        # per chunk: z1 = l1, z2 = -sg*l1 + om*l2 (same zero-input response, same transfer function H(1/z)).
        lam = astart_r[::-1].copy()    # adjoint state at each chunk *end*, section order S-1..0
        z = np.empty_like(lam)
        for i in range(S):
            k = S - 1 - i
            z[:, 2 * i] = lam[:, 2 * i]
            z[:, 2 * i + 1] = -r["sg"][k] * lam[:, 2 * i] + r["om"][k] * lam[:, 2 * i + 1]
        GX = np.zeros_like(GY)
        for n in range(L - 1, -1, -1):
            g = GY[:, n].copy()
            for i in range(S):
                k = S - 1 - i
                b0, b1, b2 = r["b"][k]
                a1, a2 = r["a"][k]
                K0, K1 = S2[k, :, n], S2[k, :, n + 1]
                if r["direct"][k]:
                    R1, R0 = K1, K0
                    if not fast:
                        acc[k, 0] += np.dot(g, S2[k, :, n + 2])
                else:
                    R1, R0 = K1 - r["sg"][k] * K0, K0                        # D[n], K[n]
                    if not fast:
                        acc[k, 0] += np.dot(g, S2[k, :, n + 2] - r["sg"][k] * K1)   # D[n + 1]
                acc[k, 1] += np.dot(g, R1); acc[k, 2] += np.dot(g, R0)
                out = b0 * g + z[:, 2 * i]
                z[:, 2 * i] = b1 * g - a1 * out + z[:, 2 * i + 1]
                z[:, 2 * i + 1] = b2 * g - a2 * out
                acc[k, 3] += np.dot(out, R1); acc[k, 4] += np.dot(out, R0)
                g = out
                if fast and i == 0:
                    acc[:, 0] += np.dot(Ulast[:, n], out)                       # T, the same number in every section's slot
            GX[:, n] = g
        gx[t * TS:(t + 1) * TS] = GX.reshape(-1)
    if scan_only:
        return acarry
    if raw:
        return gx, acc
    return (gx if tiles is not None else gx[:N],) + _normalise_grads(r, acc, fast)


def _normalise_grads(r, acc, fast=False):
    """The finalize step (csrc/sosfilt.hip finalize_section): the kernel's sums -> gradients w.r.t. the normalised coefficients."""
    S = len(r["sg"])
    om = np.where(r["direct"], 1.0, r["om"])
    sg = r["sg"]
    nf = ~r["direct"]
    lag = np.zeros((S, 5))
    lag[:, 2] = acc[:, 2]
    lag[:, 1] = np.where(nf, acc[:, 1] + sg * acc[:, 2], acc[:, 1])
    lag[:, 4] = acc[:, 4]
    lag[:, 3] = np.where(nf, acc[:, 3] + sg * acc[:, 4], acc[:, 3])
    if fast:
        pk = np.concatenate([[1.0], np.cumprod(r["d"])])[:S]                    # scale of section k's signals: 1 / pk
        rhs = acc[:, 0] * om * pk[S - 1] / pk
        lag[:, 0] = (rhs - r["b"][:, 1] * lag[:, 1] - r["b"][:, 2] * lag[:, 2]) / r["b"][:, 0]
        lag = lag * pk[:, None]
    else:
        lag[:, 0] = np.where(nf, acc[:, 0] + sg * lag[:, 1], acc[:, 0])
    gb = lag[:, :3] / om[:, None]
    ga = np.zeros((S, 3))
    ga[:, 1:] = -lag[:, 3:] / om[:, None]
    # d/da0 at a0 = 1 from scale invariance of B/A:  sum_theta theta * dL/dtheta = 0
    ga[:, 0] = -(np.sum(gb * r["b"], 1) + np.sum(ga[:, 1:] * r["a"], 1))
    return gb, ga


# ---- backward by Gram matrix (csrc/sosfilt.hip sos_bwd_gram_kernel / sos_gram_finalize_kernel) ------------------------------------------
# Inside a chunk every forward signal of the cascade is a linear function of u = (the chunk's L inputs, its 2S start-state components)
# and every adjoint signal a linear function of v = (the chunk's L adjoint inputs, the 2S components of the adjoint state entering from
# above), so each correlation sum_n g_k[n] w_k[n - j], sum_n o_k[n] w_k[n - j] over the row is <C, M> with C = sum over chunks of v u^T
# ((L + 2S) x (L + 2S)) and M built from the per-chunk basis responses of the cascade. The kernel accumulates C (fp32 products on the
# matrix cores inside a tile, fp64 across tiles), takes gx from the linear map of v, and the finalize step does the rest in fp64.

def chunk_basis_responses(r, L):
    """FW[k][n] (n = 0..L+1): w_k[n - 2] as a row vector over u = [x (L); start states (2S)];  FG[k][n], FO[k][n]: adjoint input /
    output of section k at sample n as row vectors over v = [gy (L); adjoint states (2S, adjoint section order i <-> k = S-1-i)].
    Start states enter as the kernels define them: w[-2] = s2 / om, w[-1] = s1 + (sg / om) s2;  z1 = l1, z2 = -sg l1 + om l2."""
    S = len(r["sg"]); D = L + 2 * S
    FW = np.zeros((S, L + 2, D)); FG = np.zeros((S, L, D)); FO = np.zeros((S, L, D))
    for j in range(D):
        e = np.zeros(D); e[j] = 1.0
        sig = e[:L].copy()
        for k in range(S):
            b0, b1, b2 = r["b"][k]; a1, a2 = r["a"][k]
            s1, s2 = e[L + 2 * k], e[L + 2 * k + 1]
            w2 = s2 / r["om"][k]; w1 = s1 + (r["sg"][k] / r["om"][k]) * s2
            FW[k, 0, j] = w2; FW[k, 1, j] = w1
            for n in range(L):
                w = sig[n] - a1 * w1 - a2 * w2
                sig[n] = b0 * w + b1 * w1 + b2 * w2
                FW[k, n + 2, j] = w
                w2, w1 = w1, w
        g = e[:L].copy()
        for i in range(S):
            k = S - 1 - i
            b0, b1, b2 = r["b"][k]; a1, a2 = r["a"][k]
            l1, l2 = e[L + 2 * i], e[L + 2 * i + 1]
            z1 = l1; z2 = -r["sg"][k] * l1 + r["om"][k] * l2
            for n in range(L - 1, -1, -1):
                FG[k, n, j] = g[n]
                out = b0 * g[n] + z1
                z1 = b1 * g[n] - a1 * out + z2
                z2 = b2 * g[n] - a2 * out
                FO[k, n, j] = out
                g[n] = out
    return FW, FG, FO


def gram_backward_row(r, x, gy, carries, L, fp32_tiles=False):
    """backward_row's results (gx, gb, ga) the way the Gram-matrix kernels compute them. fp32_tiles: round the scanned states to fp32 and
    form each tile's 64-chunk contribution to C in fp32 (what the matrix cores do), to measure what that costs."""
    fs = _sections_fwd(r); ads = _sections_adj(r); S = len(fs)
    G, M, P = chunk_tables(fs, L); Ga, Ma, Pa = chunk_tables(ads, L)
    TS = WAVE * L; N = len(x); nt = (N + TS - 1) // TS; D = L + 2 * S
    xp = np.zeros(nt * TS); xp[:N] = x
    gp = np.zeros(nt * TS); gp[:N] = gy
    FW, FG, FO = chunk_basis_responses(r, L)
    # gx of a chunk = (adjoint output of the last adjoint section = forward section 0) as a linear map of v: LY::YMA in the kernels
    OUT = FO[0]                                             # (L, D)
    Cm = np.zeros((D, D)); acarry = np.zeros(2 * S); gx = np.zeros(nt * TS)
    for t in range(nt - 1, -1, -1):
        X = xp[t * TS:(t + 1) * TS].reshape(WAVE, L); GY = gp[t * TS:(t + 1) * TS].reshape(WAVE, L)
        astart_r, acarry = tile_scan(GY[::-1, ::-1] @ Ga.T, Ma, Pa, acarry)
        start, _ = tile_scan(X @ G.T, M, P, carries[t])       # (the kernel reads these from what the forward pass saved)
        lam = astart_r[::-1]
        if fp32_tiles:
            start = start.astype(np.float32).astype(np.float64); lam = lam.astype(np.float32).astype(np.float64)
        U = np.concatenate([X, start], 1); V = np.concatenate([GY, lam], 1)
        Cm += (V.astype(np.float32).T @ U.astype(np.float32)).astype(np.float64) if fp32_tiles else V.T @ U
        gx[t * TS:(t + 1) * TS] = (V @ OUT.T).reshape(-1)
    gb = np.zeros((S, 3)); ga = np.zeros((S, 3))
    for k in range(S):
        Pk = FW[k] @ Cm.T                                    # P[m] = C FW[k][m]  (the finalize kernel's matrix product)
        for j in range(3):
            gb[k, j] = sum(FG[k, n] @ Pk[n + 2 - j] for n in range(L))
        for j in (1, 2):
            ga[k, j] = -sum(FO[k, n] @ Pk[n + 2 - j] for n in range(L))
        ga[k, 0] = -(np.sum(gb[k] * r["b"][k]) + np.sum(ga[k, 1:] * r["a"][k]))
    return gx[:N], gb, ga


# ---- segmented scheme for few rows (a row is one workgroup, so B*C < 512 rows leave CUs idle) --------------------
# Every row is cut into `segments` runs of tiles that are processed independently:
#   1. a scan-only pass over each segment from a zero state gives z(g), the cascade state its input alone leaves behind;
#   2. start(g + 1) = Phi_seg start(g) + z(g) with Phi_seg = Phi^(samples per segment) chains the segments (a 2S-vector recursion);
#   3. the ordinary pass runs per segment from start(g).
# The backward pass is the mirror image on the adjoint system, walking the segments downwards.
def segment_transitions(r, L, tiles_per_segment):
    n = WAVE * L * tiles_per_segment
    Phi, _ = cascade_system(_sections_fwd(r))
    Phia, _ = cascade_system(_sections_adj(r))
    return np.linalg.matrix_power(Phi, n), np.linalg.matrix_power(Phia, n)


def forward_row_segmented(r, x, L, segments):
    TS = WAVE * L
    N = len(x)
    nt = (N + TS - 1) // TS
    assert nt % segments == 0, "the model takes whole segments"
    T = nt // segments
    Phi_seg, _ = segment_transitions(r, L, T)
    z = [forward_row(r, x, L, tiles=(g * T, (g + 1) * T), scan_only=True)[0] for g in range(segments)]          # step 1
    start = [np.zeros_like(z[0])]
    for g in range(segments - 1):                                                                                # step 2
        start.append(Phi_seg @ start[g] + z[g])
    y = np.zeros(nt * TS)
    carries = np.zeros((nt, len(z[0])))
    for g in range(segments):                                                                                    # step 3
        yg, cg = forward_row(r, x, L, tiles=(g * T, (g + 1) * T), carry0=start[g])
        y[g * T * TS:(g + 1) * T * TS] = yg[g * T * TS:(g + 1) * T * TS]
        carries[g * T:(g + 1) * T] = cg[g * T:(g + 1) * T]
    return y[:N], carries


def backward_row_segmented(r, x, gy, carries, L, segments, fast=False):
    TS = WAVE * L
    N = len(x)
    nt = (N + TS - 1) // TS
    assert nt % segments == 0
    T = nt // segments
    _, Phia_seg = segment_transitions(r, L, T)
    za = [backward_row(r, x, gy, carries, L, tiles=(g * T, (g + 1) * T), scan_only=True) for g in range(segments)]
    aend = [None] * segments                         # adjoint state entering segment g from above
    aend[segments - 1] = np.zeros_like(za[0])
    for g in range(segments - 1, 0, -1):
        aend[g - 1] = Phia_seg @ aend[g] + za[g]
    gx = np.zeros(nt * TS)
    acc = 0
    for g in range(segments):
        gxg, a = backward_row(r, x, gy, carries, L, tiles=(g * T, (g + 1) * T), acarry0=aend[g], raw=True, fast=fast)
        gx[g * T * TS:(g + 1) * T * TS] = gxg[g * T * TS:(g + 1) * T * TS]
        acc = acc + a
    return (gx[:N],) + _normalise_grads(r, acc, fast)
