"""Developer timing (GPU time of the library calls, HIP events): the EQ, the compressor and the reverb at the reference's training batch
sizes, one workgroup per row / item (config.plan.*_segment = False) against the segmented default, next to the per-sample time of the full batch."""
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


def gpu_ms(fn, B, C, N, ranges, env):
    config.plan.sos_segment = config.plan.dyn_segment = env != "0"
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
    return sum(sum(v) for v in t.values()) / 30


for name, fn, ranges, shapes, full in (("parametric_eq", D.parametric_eq, R, ((8, 2, 131072), (16, 2, 131072), (32, 2, 131072)), (256, 2, 131072)),
                                       ("compressor", D.compressor, CR, ((8, 2, 262144), (16, 2, 262144), (32, 2, 262144)), (256, 2, 262144))):
    tf = gpu_ms(fn, *full, ranges, "auto")
    per = tf / (full[0] * full[1] * full[2])
    print(f"{name} {full}: {tf:.4f} ms = {per * 1e9:.3f} ps per channel-sample")
    for shp in shapes:
        tp, ts = gpu_ms(fn, *shp, ranges, "0"), gpu_ms(fn, *shp, ranges, "auto")
        n = shp[0] * shp[1] * shp[2]
        print(f"  {shp}: one workgroup per row/item {tp:.4f} ms, segmented {ts:.4f} ms = {ts / n / per:.2f} x the full batch's per-sample time", flush=True)
