"""StyleTransferChain fwd+bwd step (EQ -> compressor -> reverb -> gain with gradients) with the fused training forward on / off
(config.plan.chain_fused_grad), as replayed graphs, blocks interleaved.   usage: python scripts/chain_step_ab.py B C N [B C N ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd.chain import StyleTransferChain

SR, dev = 44100, "cuda"
args = [int(v) for v in sys.argv[1:]] or [256, 2, 131072]
for B, C, N in zip(args[0::3], args[1::3], args[2::3]):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand(B, C, N, device=dev, generator=g) * 2 - 1
    w = torch.randn(B, 2, N, device=dev, generator=g)
    ps = [torch.rand(B, k, device=dev, generator=g).requires_grad_(True) for k in (18, 6, 25, 1)]
    chain = StyleTransferChain(SR, noise_seed=7)
    for proc in (chain.equalizer, chain.compressor, chain.reverb, chain.gain):
        proc.validate_range = False

    def step():
        for p in ps:
            p.grad = None
        chain.process_normalized(x, *ps).backward(w)
    graphs = {}
    for name, mode in (("two_launches", False), ("fused", True)):
        D.config.plan.chain_fused_grad = mode
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            step()
        graphs[name] = gph
    D.config.plan.chain_fused_grad = None
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        for gph in graphs.values():
            gph.replay()
        torch.cuda.synchronize()
    times = {k: [] for k in graphs}
    reps = 20 if B >= 128 else 100
    for blk in range(6):
        for name in (list(graphs) if blk % 2 == 0 else list(graphs)[::-1]):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                graphs[name].replay()
            torch.cuda.synchronize()
            times[name].append((time.perf_counter() - t0) / reps * 1e3)
    print(json.dumps({"shape": [B, C, N], **{k + "_ms": round(float(np.median(v)), 4) for k, v in times.items()}}))
    del graphs
