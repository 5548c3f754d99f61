"""Developer timing: the reference's chain step (16, 1, 131072) as one replayed HIP graph + per-stage graph steps (run from any checkout of the tree)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dasp_pytorch_amd as D
from bench import graph_step_ms, SR
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
off = torch.zeros(1, dtype=torch.int64, device=dev)
chain = D.chain.StyleTransferChain(SR, device_noise=True, noise_seed=7, noise_seed_offset=off)
xc = rnd(16, 1, 131072) * 2 - 1
pcs = [(rnd(16, n) * 0.9 + 0.05).requires_grad_(True) for n in chain.num_params]
wc = torch.randn(16, 2, 131072, device=dev, generator=g)
out = {}
def chain_step():
    for p in pcs: p.grad = None
    chain.process_normalized(xc, *pcs).backward(wc)
out["chain"] = round(graph_step_ms(chain_step, replays=100, blocks=3, ramp_s=0.5), 4)
x2 = (rnd(16, 2, 131072) * 2 - 1)
w2 = torch.randn(16, 2, 131072, device=dev, generator=g)
rv = [rnd(16).requires_grad_(True) for _ in range(25)]
def rev_step():
    for p in rv: p.grad = None
    D.noise_shaped_reverberation(x2, SR, *rv, noise_seed=7).backward(w2)
out["reverb (16,2,131072) no gx"] = round(graph_step_ms(rev_step, replays=100, blocks=3, ramp_s=0.3), 4)
x1 = (rnd(16, 1, 131072) * 2 - 1)
def rev_step1():
    for p in rv: p.grad = None
    D.noise_shaped_reverberation(x1.requires_grad_(True), SR, *rv, noise_seed=7).backward(w2)
out["reverb (16,1,131072) gx"] = round(graph_step_ms(rev_step1, replays=100, blocks=3, ramp_s=0.3), 4)
print(json.dumps(out))
