"""GPU time of the fused EQ -> compressor forward (dasp_chain_forward) against the two separate forward calls, at the reference's target-synthesis
shape (16, 1, 262144) (examples/style_transfer.py:293-299) and at (256, 2, 131072). HIP events around the C entry points (sum per step) and wall
time per step of the Python calls. usage: python scripts/chain_fwd_time.py [out.json]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D  # noqa: E402
from dasp_pytorch_amd import _lib, ops  # noqa: E402
from dasp_pytorch_amd.functional import _PEQ_TYPES  # noqa: E402

SR = 44100
PEQ = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
       (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
DYN = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]


def bench(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    _lib.timers.start(every=1)
    for _ in range(20):
        fn()
    kt = _lib.timers.stop()
    return wall * 1e3, sum(sum(v) for v in kt.values()) / 20, {k: round(sum(v) / 20, 4) for k, v in kt.items()}


def main():
    out = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    t_end = time.perf_counter() + 1.0
    a = torch.rand(1 << 26, device="cuda")
    while time.perf_counter() < t_end:      # clock ramp
        a.mul_(1.0001)
    torch.cuda.synchronize()
    for B, C, N in ((16, 1, 262144), (256, 2, 131072), (32, 2, 131072), (8, 1, 262144)):
        x = torch.rand(B, C, N, device="cuda", generator=g) * 2 - 1
        pn = torch.rand(B, 18, device="cuda", generator=g)
        cu = torch.rand(B, 6, device="cuda", generator=g)
        lo = torch.tensor([r[0] for r in DYN], device="cuda"); hi = torch.tensor([r[1] for r in DYN], device="cuda")
        comp = cu * (hi - lo) + lo
        ctl = torch.cat([comp[:, :3], comp[:, 4:]], 1).contiguous()
        elo = [float(r[0]) for r in PEQ]; espan = [float(r[1] - r[0]) for r in PEQ]
        eq = D.ParametricEQ(SR)
        eq.validate_range = False
        cols = [comp[:, i].contiguous() for i in range(6)]
        with torch.no_grad():
            sep = bench(lambda: D.compressor(eq.process_normalized(x, pn), SR, *cols))
            fus = bench(lambda: ops.chain_eq_compressor_forward(x, pn, _PEQ_TYPES, elo, espan, float(SR), ctl))
        out[f"({B},{C},{N})"] = {"separate": {"wall_ms": round(sep[0], 4), "gpu_ms": round(sep[1], 4), "calls": sep[2]},
                                 "fused": {"wall_ms": round(fus[0], 4), "gpu_ms": round(fus[1], 4), "calls": fus[2]},
                                 "gpu_ratio": round(fus[1] / sep[1], 3)}
        print(f"({B},{C},{N}): separate gpu {sep[1]:.4f} ms wall {sep[0]:.4f} | fused gpu {fus[1]:.4f} ms wall {fus[0]:.4f} | ratio {fus[1] / sep[1]:.3f}")
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
