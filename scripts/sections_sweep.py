"""Developer timing: an 8-section cascade as one S=8 launch per direction (one wave per SIMD in the backward kernel) against two
chained launches (4 + 4, 6 + 2 sections): GPU time of the library calls, forward + backward with gradients for sos and x."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import _lib
from dasp_pytorch_amd.ops import SosFiltFunction
SR = 44100
g = torch.Generator(device="cuda:0").manual_seed(0)
rnd = lambda *s: torch.rand(*s, device="cuda:0", generator=g)


def make_sos(B, S):
    rows = []
    for k in range(S):
        gain = rnd(B) * 24 - 12; fc = rnd(B) * 8000 + 40 * (k + 1); q = rnd(B) * 4 + 0.3
        b, a = D.signal.biquad(gain, fc, q, SR, "peaking")
        rows.append(torch.cat([b, a], -1))
    return torch.stack(rows, 1).contiguous()


def gpu_ms(split, B, C, N, S):
    g.manual_seed(S)                                      # the same filters and signals for every split of one cascade
    sos = make_sos(B, S).requires_grad_(True)
    x = (rnd(B, C, N) * 2 - 1).requires_grad_(True)
    w = torch.randn(B, C, N, device="cuda:0", generator=g)
    def step():
        x.grad = None; sos.grad = None
        y, s0 = x, 0
        for n in split:
            y = SosFiltFunction.apply(sos[:, s0:s0 + n], y); s0 += n
        y.backward(w)
        return y
    for _ in range(10): step()
    torch.cuda.synchronize()
    _lib.timers.start(every=1)
    for _ in range(20): step()
    t = _lib.timers.stop()
    y = step().detach(); gs = sos.grad.clone(); gx = x.grad.clone()
    return sum(sum(v) for v in t.values()) / 20, y, gs, gx


for shp in ((256, 2, 131072), (16, 2, 131072)):
    for S, splits in ((8, ((8,), (4, 4), (6, 2))), (7, ((7,), (4, 3), (6, 1))), (12, ((8, 4), (6, 6), (4, 4, 4)))):
        base = None
        for split in splits:
            ms, y, gs, gx = gpu_ms(split, *shp, S)
            if base is None:
                base = (y, gs, gx)
            err = [float((a - b).abs().max() / b.abs().max()) for a, b in zip((y, gs, gx), base)]
            print(shp, "S", S, "split", split, "gpu %.4f ms" % ms, "vs first: y %.1e gsos %.1e gx %.1e" % tuple(err), flush=True)
