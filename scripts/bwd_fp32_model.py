"""Developer study (CPU only): where the fp32 error of the EQ parameter gradients comes from, and what each remedy buys.

Emulates the arithmetic of sos_bwd_kernel for one row in numpy float32 (fused multiply-adds emulated through float64), vectorised over
the row's 16-sample chunks: every chunk starts from its exact (fp64 truth, rounded to fp32) forward state and costate, exactly as the
kernel restarts every chunk from the lane scans, recomputes the forward sections (direct form II where the prep kernel allows it, normal
form otherwise; optionally in the monic form, feed-through 1), runs the adjoint sections in transposed direct form II and forms the
coefficient correlations per section. The correlations are accumulated in several ways and pushed through the fp64 RBJ Jacobian.
Printed: the error of the 18 control gradients per item in the tests' metric (L-inf / peak over the item's controls) per variant.

    N=131072 NRAND=10 python scripts/bwd_fp32_model.py
"""
import os
import sys

import numpy as np
import scipy.signal as sg

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dasp_oracle as orc          # noqa: E402
from oracle.chunkscan_model import realize     # noqa: E402

SR, L, WAVE, W = 44100, 16, 64, 4
F = np.float32
D = np.float64
R = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
     (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]


def fma(a, b, c):
    """fp32 fused multiply-add: the product of two fp32 numbers is exact in fp64."""
    return (np.asarray(a, D) * np.asarray(b, D) + np.asarray(c, D)).astype(F)


def design_jacobian(p):
    """d(b0,b1,b2,a1,a2 normalised)/d(gain, fc, q) per section by central differences in fp64: (S, 5, 3)."""
    S = 6
    J = np.zeros((S, 5, 3))
    for k in range(S):
        for d in range(3):
            h = 1e-6 * max(1.0, abs(p[3 * k + d]))
            out = []
            for sgn in (+1, -1):
                q = p.astype(D).copy()
                q[3 * k + d] += sgn * h
                sos = orc.peq_sos(q[None], SR)[0, k]
                out.append(np.array([sos[0], sos[1], sos[2], sos[4], sos[5]]) / sos[3])
            J[k, :, d] = (out[0] - out[1]) / (2 * h)
    return J


def truth(sos, x, gy):
    """fp64 whole-row signals: per section all-pole signal w, adjoint input g and output o."""
    S = sos.shape[0]
    u = x.astype(D)
    Wsig = []
    for k in range(S):
        b, a = sos[k, :3] / sos[k, 3], sos[k, 3:] / sos[k, 3]
        w = sg.lfilter([1.0], a, u)
        Wsig.append(w)
        u = sg.lfilter(b, [1.0], w)
    y = u
    g = gy.astype(D)
    Gs, Os = [None] * S, [None] * S
    for k in range(S - 1, -1, -1):
        b, a = sos[k, :3] / sos[k, 3], sos[k, 3:] / sos[k, 3]
        o = sg.lfilter(b, a, g[::-1])[::-1]
        Gs[k], Os[k] = g, o
        g = o
    return y, g, Wsig, Gs, Os


def emulate(p, x, gy, monic=False):
    sos = orc.peq_sos(p.astype(D)[None], SR)[0]
    r = realize(sos)
    S, N = 6, len(x)
    nch = N // L
    nt = nch // WAVE
    y, gx, Wsig, Gs, Os = truth(sos, x, gy)

    def lag(w, i):
        return np.concatenate([np.zeros(i), w[:N - i]]) if i else w
    gb_true = np.array([[np.dot(Gs[k], lag(Wsig[k], i)) for i in range(3)] for k in range(S)])
    ga_true = np.array([[-np.dot(Os[k], lag(Wsig[k], i)) for i in (1, 2)] for k in range(S)])
    J = design_jacobian(p)

    def to_params(gb, ga):
        return np.einsum("kc,kcd->kd", np.concatenate([gb, ga], 1), J).reshape(-1)
    gp_true = to_params(gb_true, ga_true)

    # ---- forward recomputation in fp32, vectorised over chunks ----------------------------------------------------------------------
    ucur = x.astype(F).reshape(nch, L).copy()
    S2 = np.zeros((S, nch, L + 2), F)
    Pk = np.concatenate([[1.0], np.cumprod(r["d"])])     # Pk[k] = prod_{j<k} b0_j: scale of section k's input in the monic recomputation
    u_last = None
    for k in range(S):
        b0, b1, b2 = r["b"][k]
        a1, a2 = r["a"][k]
        w = Wsig[k] / (Pk[k] if monic else 1.0)
        if monic:
            b0, b1, b2 = 1.0, b1 / b0, b2 / b0
        b0, b1, b2, na1, na2 = F(b0), F(b1), F(b2), F(-a1), F(-a2)
        if k == S - 1:
            u_last = ucur.copy()
        wm1 = np.concatenate([[0.0], w[:-1]])[::L][:nch]
        wm2 = np.concatenate([[0.0, 0.0], w[:-2]])[::L][:nch]
        if r["direct"][k]:
            w2, w1 = wm2.astype(F), wm1.astype(F)
            for n in range(L):
                uu = ucur[:, n]
                S2[k, :, n] = w2
                wn = fma(na1, w1, fma(na2, w2, uu))
                ucur[:, n] = fma(b1, w1, fma(b2, w2, wn)) if monic else fma(b0, wn, fma(b1, w1, b2 * w2))
                w2, w1 = w1, wn
            S2[k, :, L], S2[k, :, L + 1] = w2, w1
        else:
            sgk, omk, nk = F(r["sg"][k]), F(r["om"][k]), F(-r["kom"][k])
            d = r["d"][k]
            g1, g2 = (F(r["g1"][k] / d), F(r["g2"][k] / d)) if monic else (F(r["g1"][k]), F(r["g2"][k]))
            s2 = (r["om"][k] * wm2).astype(F)
            s1 = (wm1 - r["sg"][k] * wm2).astype(F)
            for n in range(L):
                uu = ucur[:, n].copy()
                S2[k, :, n] = s2
                ucur[:, n] = fma(g1, s1, fma(g2, s2, uu)) if monic else fma(g1, s1, fma(g2, s2, F(d) * uu))
                t1 = fma(sgk, s1, fma(nk, s2, uu))
                s2 = fma(omk, s1, sgk * s2)
                s1 = t1
            S2[k, :, L] = s2
            S2[k, :, L + 1] = fma(omk, s1, sgk * s2)
    # ---- adjoint sections (transposed direct form II) and the factors of the correlations ---------------------------------------------
    gcur = gy.astype(F).reshape(nch, L).copy()
    fa = np.zeros((S, 5, nch, L), F)          # left factor of each correlation product (g for b0 b1 b2, o for a1 a2)
    o_last = None
    for k in range(S - 1, -1, -1):
        b0, b1, b2 = (F(v) for v in r["b"][k])
        na1, na2 = F(-r["a"][k][0]), F(-r["a"][k][1])
        o, g = Os[k], Gs[k]
        op1 = np.concatenate([o[1:], [0.0]]); gp1 = np.concatenate([g[1:], [0.0]])
        z2 = (r["b"][k][2] * gp1 - r["a"][k][1] * op1)[L - 1::L][:nch].astype(F)
        z1 = (o - r["b"][k][0] * g)[L - 1::L][:nch].astype(F)
        for n in range(L - 1, -1, -1):
            gg = gcur[:, n].copy()
            fa[k, 0, :, n] = fa[k, 1, :, n] = fa[k, 2, :, n] = gg
            oo = fma(b0, gg, z1)
            z1 = fma(b1, gg, fma(na1, oo, z2))
            z2 = fma(b2, gg, na2 * oo)
            fa[k, 3, :, n] = fa[k, 4, :, n] = oo
            gcur[:, n] = oo
        if k == S - 1:
            o_last = gcur.copy()
    shift = (2, 1, 0, 1, 0)                    # right factor: S2[n + shift]

    om = np.where(r["direct"], 1.0, r["om"])

    def finish(acc):      # acc (S, 5) float64 sums -> control gradients
        if monic:
            acc = acc * Pk[:S, None]
        return to_params(acc[:, :3] / om[:, None], -acc[:, 3:] / om[:, None])

    A = fa.reshape(S, 5, nt, WAVE, L)
    B = np.stack([np.stack([S2[k, :, s:s + L] for s in shift]) for k in range(S)]).reshape(S, 5, nt, WAVE, L)

    def lane_sums(mode, A=A, B=B):
        """mode 'run32': the kernel before this study, one running fp32 sum per lane over all its tiles; 'tile32': per-tile fp32 sums
        added to a running fp32 sum; 'tile64': per-tile fp32 sums added to a running fp64 sum. Then lanes (fp32 butterfly for the fp32
        running sums), then waves in fp64."""
        lead = A.shape[:-3]
        tot = np.zeros(lead)
        for wv in range(W):
            tiles = [t for t in range(nt - 1, -1, -1) if (nt - 1 - t) % W == wv]
            acc = np.zeros(lead + (WAVE,), D if mode == "tile64" else F)
            for t in tiles:
                if mode == "run32":
                    for n in range(L - 1, -1, -1):
                        acc = fma(A[..., t, :, n], B[..., t, :, n], acc)
                else:
                    tacc = np.zeros(lead + (WAVE,), F)
                    for n in range(L - 1, -1, -1):
                        tacc = fma(A[..., t, :, n], B[..., t, :, n], tacc)
                    acc = acc + tacc
            if mode == "tile64":
                tot += acc.sum(-1)
            else:
                v = acc
                for dsh in (32, 16, 8, 4, 2, 1):
                    v = v[..., :dsh] + v[..., dsh:2 * dsh]
                tot += v[..., 0].astype(D)
        return tot

    res = {}
    res["signals only (fp64 sums)"] = finish((A.astype(D) * B.astype(D)).sum((2, 3, 4)))
    res["running fp32 sums (r01 kernel)"] = finish(lane_sums("run32"))
    for mode in ("tile32", "tile64"):
        acc = lane_sums(mode)
        res[f"{mode}"] = finish(acc)
        # b0 correlation dropped: sum_i b_i gb_i = T = <g_k, y_k>, the same for every section k (adjoint identity over the whole row);
        # T is taken between the last two sections: T = <o_last, u_last> (both are in registers there)
        T = lane_sums(mode, o_last.reshape(nt, WAVE, L), u_last.reshape(nt, WAVE, L))
        acc3 = acc.copy()
        for k in range(S):
            bb = r["b"][k]
            Tk = T * (Pk[S - 1] / Pk[k] if monic else 1.0)      # in the units of this section's (scaled) sums
            acc3[k, 0] = (Tk * om[k] - bb[1] * acc[k, 1] - bb[2] * acc[k, 2]) / bb[0]
        res[f"{mode} + identity"] = finish(acc3)
    peak = np.abs(gp_true).max()
    return {name: np.abs(gp - gp_true).max() / peak for name, gp in res.items()}


def main():
    rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
    N = int(os.environ.get("N", 65536))
    nrand = int(os.environ.get("NRAND", 2))
    lo = np.array([r[0] for r in R]); hi = np.array([r[1] for r in R])
    p = (rng.random((6 + nrand, 18)) * (hi - lo) + lo).astype(np.float32)
    for b in range(4):            # the corners of scripts/eq_accuracy.py
        p[b, 1::3] = lo[1::3]; p[b, 2::3] = hi[2::3]; p[b, 0::3] = 20.0 if b % 2 else -20.0
    p[2, 2::3] = lo[2::3]; p[3, 2::3] = lo[2::3]
    for b in (4, 5):
        p[b, 1::3] = hi[1::3]; p[b, 2::3] = hi[2::3] if b == 4 else lo[2::3]; p[b, 0::3] = 20.0 if b % 2 else -20.0
    names = None
    worst = {}
    for b in range(len(p)):
        x = (rng.random(N) * 2 - 1).astype(np.float32)
        w = rng.standard_normal(N).astype(np.float32)
        for monic in (False, True):
            e = emulate(p[b], x, w, monic=monic)
            if names is None:
                names = list(e)
                for i, k in enumerate(names):
                    print(f"  col {i}: {k}")
            print(f"item {b:2d} {'monic ' if monic else 'as is '}" + "  ".join(f"{v:.2e}" for v in e.values()))
            for k, v in e.items():
                worst[(monic, k)] = max(worst.get((monic, k), 0), v)
    for monic in (False, True):
        print(f"worst   {'monic ' if monic else 'as is '}" + "  ".join(f"{worst[(monic, k)]:.2e}" for k in names))


if __name__ == "__main__":
    main()
