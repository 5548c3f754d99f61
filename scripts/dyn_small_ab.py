"""Developer timing: compressor fwd+bwd graph replays at the reference's training batch sizes (segmented items); the library under test is
DASP_HIP_LIB (scripts/build_variant_one.sh) or the in-tree one."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from bench import graph_step_ms, SR
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
RANGES = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]
out = {"lib": os.environ.get("DASP_HIP_LIB", "in-tree").split("/")[-2] if os.environ.get("DASP_HIP_LIB") else "in-tree"}
for B, C, N in ((8, 2, 262144), (16, 2, 262144), (32, 2, 262144), (16, 1, 131072), (8, 2, 131072)):
    ctl = [(rnd(B) * (hi - lo) + lo).requires_grad_(True) for lo, hi in RANGES]
    x = (rnd(B, C, N) * 2 - 1).requires_grad_(True)
    w = torch.randn(B, C, N, device=dev, generator=g)
    def step():
        x.grad = None
        for c in ctl:
            c.grad = None
        D.compressor(x, SR, *ctl).backward(w)
    out[f"({B},{C},{N})"] = round(graph_step_ms(step, replays=200, blocks=3, ramp_s=0.3), 4)
print(json.dumps(out))
