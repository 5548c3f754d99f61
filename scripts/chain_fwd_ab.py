"""Fused EQ -> compressor forward (csrc/chainfwd.hip), GPU-bound time as a replayed graph. usage: [DASP_HIP_LIB=...] python scripts/chain_fwd_ab.py tag"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import ops
from dasp_pytorch_amd.functional import _PEQ_TYPES
from bench import graph_step_ms, PEQ_RANGES, SR
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(3)
rng = [(-60, 0), (1, 20), (5, 100), (1e-3, 12), (0, 12)]
out = {"tag": sys.argv[1] if len(sys.argv) > 1 else ""}
elo = [float(r[0]) for r in PEQ_RANGES]; espan = [float(r[1] - r[0]) for r in PEQ_RANGES]
for B, C, N in ((256, 2, 131072), (16, 1, 262144)):
    x = torch.rand(B, C, N, device=dev, generator=g) * 2 - 1
    pn = torch.rand(B, 18, device=dev, generator=g)
    ctl = torch.stack([torch.rand(B, device=dev, generator=g) * (hi - lo) + lo for lo, hi in rng], 1).contiguous()
    def step():
        with torch.no_grad():
            ops.chain_eq_compressor_forward(x, pn, _PEQ_TYPES, elo, espan, float(SR), ctl)
    out[f"({B},{C},{N})"] = graph_step_ms(step, replays=50, blocks=5, ramp_s=0.3)
print(json.dumps(out))
