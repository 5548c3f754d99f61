"""GPU-bound time per fwd+bwd step of the ops at the reference's training batch sizes, box-independently: the whole step (forward,
backward, gradient accumulation) is captured ONCE into a HIP graph and replayed back to back - a replay costs the host ~15 us, so the
queue never runs dry and no event pair sits between kernels (bench.py's `secondary` used HIP events around every library call of an
eager, host-bound loop: the idle gaps let power-limited kernels clock up, by a different amount on every box).
usage: [DASP_HIP_LIB=tools/<variant>/libdasp_hip.so] python scripts/small_batch_graph.py [tag]   -> one JSON line"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from bench import graph_step_ms, PEQ_RANGES, SR

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
out = {"tag": sys.argv[1] if len(sys.argv) > 1 else "", "lib": os.environ.get("DASP_HIP_LIB", "in-tree")}
rng = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]


def op(name, B, C, N, ctl, call, oc=None):
    x = (rnd(B, C, N) * 2 - 1).requires_grad_(True)
    w = torch.randn(B, oc or C, N, device=dev, generator=g)

    def step():
        x.grad = None
        for c in ctl:
            c.grad = None
        call(x, ctl).backward(w)
    out[name] = graph_step_ms(step)


cols = lambda B, r: [(rnd(B) * (hi - lo) + lo).requires_grad_(True) for lo, hi in r]
op("parametric_eq_b16", 16, 2, 131072, cols(16, PEQ_RANGES), lambda x, c: D.parametric_eq(x, SR, *c))
op("parametric_eq_b256", 256, 2, 131072, cols(256, PEQ_RANGES), lambda x, c: D.parametric_eq(x, SR, *c))
op("compressor_b8", 8, 2, 262144, cols(8, rng), lambda x, c: D.compressor(x, SR, *c))
op("reverb_b8", 8, 2, 131072, cols(8, [(0, 1)] * 25), lambda x, c: D.noise_shaped_reverberation(x, SR, *c, device_noise=True, noise_seed=5), oc=2)
op("reverb_b128", 128, 2, 262144, cols(128, [(0, 1)] * 25), lambda x, c: D.noise_shaped_reverberation(x, SR, *c, device_noise=True, noise_seed=5), oc=2)
off = torch.zeros(1, dtype=torch.int64, device=dev)
chain = D.chain.StyleTransferChain(SR, device_noise=True, noise_seed=7, noise_seed_offset=off)
xc = rnd(16, 1, 131072) * 2 - 1
pcs = [(rnd(16, n) * 0.9 + 0.05).requires_grad_(True) for n in chain.num_params]
wc = torch.randn(16, 2, 131072, device=dev, generator=g)


def chain_step():
    for p in pcs:
        p.grad = None
    chain.process_normalized(xc, *pcs).backward(wc)
out["style_transfer_chain_b16"] = graph_step_ms(chain_step)
print(json.dumps(out))
