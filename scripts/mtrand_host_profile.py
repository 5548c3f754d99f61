"""Host-side cost of one device draw of torch's CPU random stream (dasp_pytorch_amd._mt19937.randn_cpu_stream): wall time per call with
the GPU work small (so that the host is what is timed), and a cProfile of 300 calls.  usage: python scripts/mtrand_host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dasp_pytorch_amd import _mt19937 as mt

dev = "cuda:0"
size = (16, 12, 66558)
torch.manual_seed(0)
mt.randn_cpu_stream(*size, device=dev)
torch.cuda.synchronize()


def once(defer):
    if defer:
        t, p = mt.randn_cpu_stream(*size, device=dev, defer=True)
        p.finish()
    else:
        t = mt.randn_cpu_stream(*size, device=dev)
    return t


for defer in (False, True):
    for _ in range(20):
        once(defer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        once(defer)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"defer={defer}: {(t1 - t0) / 300 * 1e6:.1f} us of wall time per call (host issue + the state's read-back; GPU work per call ~107 us)")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    once(True)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
