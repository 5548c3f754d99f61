"""Randomized check of the device reproduction of torch's CPU random stream (csrc/mtrand.hip) against torch.randn itself: random seeds,
positions in the stream and sizes - small ones, sizes around the unit (159,744 values), chunk-stride and giant-jump borders, a few
large ones - values within 1e-6 of the largest sample, the generator state afterwards bit-equal, the next CPU draws equal.
usage: python scripts/fuzz_mtrand.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dasp_pytorch_amd import _mt19937 as mt

dev = "cuda:0"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
U = 159744
t0 = time.time()
n_cases, worst, fails, biggest = 0, 0.0, 0, 0
while time.time() - t0 < budget:
    kind = rng.integers(0, 10)
    if kind < 4:
        n = int(rng.integers(16, 5000))
    elif kind < 7:
        n = int(rng.integers(1, 40)) * U + int(rng.integers(-700, 700))
    elif kind < 9:
        n = int(rng.choice([256, 257, 300, 512, 513, 600])) * U + int(rng.integers(-2000, 2000))       # chunk strides 1 / 2 / 3, giant jump
    else:
        n = int(rng.integers(700, 1300)) * U + int(rng.integers(0, 5000))                              # strides 3 .. 6, several giant jumps
    n = max(n, 16)
    seed, burn = int(rng.integers(0, 2 ** 31)), int(rng.choice([0, 1, 7, 100, 623, 624, 625, int(rng.integers(0, 3000))]))
    torch.manual_seed(seed)
    if burn:
        torch.rand(burn)
    s0 = torch.get_rng_state()
    ref = torch.randn(n)
    s_ref = torch.get_rng_state()
    after_ref = torch.rand(3)
    torch.set_rng_state(s0)
    got = mt.randn_cpu_stream(n, device=dev)
    s_got = torch.get_rng_state()
    after_got = torch.rand(3)
    err = float((got.cpu() - ref).abs().max() / ref.abs().max())
    ok = err <= 1e-6 and torch.equal(s_got, s_ref) and torch.equal(after_got, after_ref)
    if not ok:
        fails += 1
        print("FAIL", dict(n=n, seed=seed, burn=burn, err=err, state=bool(torch.equal(s_got, s_ref))), flush=True)
    worst, biggest, n_cases = max(worst, err), max(biggest, n), n_cases + 1
print(f"fuzz_mtrand: {n_cases} draws in {time.time() - t0:.0f} s, largest {biggest} values, worst |device - torch.randn| / max|.| = {worst:.3g}, "
      f"generator state and following draws equal in all but {fails}: {'OK' if fails == 0 else 'FAILED'}")
