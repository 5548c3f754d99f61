"""Developer timing: EQ fwd+bwd graph replays at the reference's training batch sizes (segmented rows); run with DASP_SEG_GRAM=0 / 1."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from bench import graph_step_ms, PEQ_RANGES, SR
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
out = {"what": "parametric_eq fwd+bwd, graph step ms"}
for B, C, N, gx in ((8, 2, 131072, True), (16, 2, 131072, True), (16, 1, 131072, False), (32, 2, 131072, True), (16, 2, 262144, True)):
    cols = [(rnd(B) * (hi - lo) + lo).requires_grad_(True) for lo, hi in PEQ_RANGES]
    x = (rnd(B, C, N) * 2 - 1).requires_grad_(gx)
    w = torch.randn(B, C, N, device=dev, generator=g)
    def step():
        x.grad = None
        for c in cols:
            c.grad = None
        D.parametric_eq(x, SR, *cols).backward(w)
    out[f"({B},{C},{N}){'' if gx else ' no gx'}"] = round(graph_step_ms(step, replays=200, blocks=3, ramp_s=0.3), 4)
print(json.dumps(out))
