"""Developer timing: segment length sweep of the segmented EQ / compressor at small batches (GPU time of the library calls)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import _lib, config
SR = 44100
R = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
     (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
CR = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]
g = torch.Generator(device="cuda:0").manual_seed(0)
rnd = lambda *s: torch.rand(*s, device="cuda:0", generator=g)


def gpu_ms(fn, B, C, N, ranges):
    x = (rnd(B, C, N) * 2 - 1).requires_grad_(True)
    cols = [(rnd(B) * (hi - lo) + lo).requires_grad_(True) for lo, hi in ranges]
    w = torch.randn(B, C, N, device="cuda:0", generator=g)
    def step():
        x.grad = None
        for c in cols: c.grad = None
        fn(x, SR, *cols).backward(w)
    for _ in range(30): step()
    torch.cuda.synchronize()
    _lib.timers.start(every=1)
    for _ in range(30): step()
    t = _lib.timers.stop()
    return {k.replace("dasp_", ""): round(sum(v) / 30, 4) for k, v in t.items()}


for name, fn, ranges, key, shapes, tiles in (("parametric_eq", D.parametric_eq, R, "sos_segment_tiles", ((8, 2, 131072), (16, 2, 131072), (32, 2, 131072)), (8, 16, 32, 64)),
                                             ("compressor", D.compressor, CR, "dyn_segment_tiles", ((8, 2, 262144), (16, 2, 262144), (32, 2, 262144)), (16, 32, 64, 128))):
    for shp in shapes:
        for t in tiles:
            setattr(config.plan, key, t)
            r = gpu_ms(fn, *shp, ranges)
            print(name, shp, "tiles per segment", t, "total %.4f ms" % sum(r.values()), r, flush=True)
        setattr(config.plan, key, None)
        r = gpu_ms(fn, *shp, ranges)
        print(name, shp, "planner", "total %.4f ms" % sum(r.values()), flush=True)
