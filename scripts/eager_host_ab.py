"""Eager wall time per fwd+bwd step of the small-batch ops (host-bound: op dispatch + autograd around 50 - 90 us of kernels), for one tree.
usage: PYTHONPATH=<tree> python scripts/eager_host_ab.py <label>   (scripts/gpu_r6_host_ab.sh runs the round-5 tree and this tree alternately)"""
import json
import sys
import time

import numpy as np
import torch

import dasp_pytorch_amd as D

SR, dev = 44100, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
PEQ = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
       (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
DYN = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]


def timed(fn, x, w, ctl):
    def step():
        x.grad = None
        for c in ctl:
            c.grad = None
        fn(x, SR, *ctl).backward(w)
    for _ in range(20):
        step()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 50 * 1e3)
    return round(float(np.median(ts)), 4)


out = {"tree": sys.argv[1] if len(sys.argv) > 1 else "?", "file": D.__file__.split("/")[-3]}
x = (rnd(16, 2, 131072) * 2 - 1).requires_grad_(True); w = torch.randn(16, 2, 131072, device=dev, generator=g)
out["parametric_eq (16,2,131072)"] = timed(D.parametric_eq, x, w, [(rnd(16) * (hi - lo) + lo).requires_grad_(True) for lo, hi in PEQ])
x = (rnd(8, 2, 262144) * 2 - 1).requires_grad_(True); w = torch.randn(8, 2, 262144, device=dev, generator=g)
out["compressor (8,2,262144)"] = timed(D.compressor, x, w, [(rnd(8) * (hi - lo) + lo).requires_grad_(True) for lo, hi in DYN])
x = (rnd(8, 2, 131072) * 2 - 1).requires_grad_(True); w = torch.randn(8, 2, 131072, device=dev, generator=g)
out["reverb device_noise (8,2,131072)"] = timed(lambda xx, sr, *c: D.noise_shaped_reverberation(xx, sr, *c, device_noise=True), x, w, [rnd(8).requires_grad_(True) for _ in range(25)])
print(json.dumps(out))
