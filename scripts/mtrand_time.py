"""Time the device reproduction of torch's CPU random stream (csrc/mtrand.hip) against the host draw it replaces, at the reverb's noise
shapes. usage: python scripts/mtrand_time.py [bs ...]   (noise tensor = (2 bs, 12, 65536 + 1022))"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dasp_pytorch_amd import _mt19937 as mt

dev = "cuda:0"
for bs in [int(a) for a in sys.argv[1:]] or [8, 128]:
    size = (2 * bs, 12, 66558)
    torch.manual_seed(0)
    mt.randn_cpu_stream(*size, device=dev)          # warm-up: table upload, self-check
    torch.cuda.synchronize()
    walls, gpus = [], []
    for _ in range(5):
        torch.manual_seed(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        out = mt.randn_cpu_stream(*size, device=dev)
        e1.record()
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
        gpus.append(e0.elapsed_time(e1))
    torch.manual_seed(1)
    t0 = time.perf_counter()
    ref = torch.randn(*size)
    t_host = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    ref_d = ref.to(dev)
    torch.cuda.synchronize()
    t_copy = (time.perf_counter() - t0) * 1e3
    err = float((out - ref_d).abs().max())
    print(json.dumps({"noise": list(size), "values": out.numel(), "device_wall_ms": sorted(walls)[2], "device_gpu_ms": sorted(gpus)[2],
                      "host_randn_ms": t_host, "host_copy_ms": t_copy, "max_abs_diff": err}))
