"""Developer timing: where segmented rows stop paying against one workgroup per row (whose backward is the Gram-matrix kernel since round 4):
EQ fwd+bwd as a replayed HIP graph at (B, 2, 131072) for B = 8 .. 96, DASP_SOS_SEGMENT=0 against the library's rule."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from bench import graph_step_ms, PEQ_RANGES, SR

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
for B in (8, 16, 24, 32, 40, 48, 56, 63, 64, 96):
    cols = [(rnd(B) * (hi - lo) + lo).requires_grad_(True) for lo, hi in PEQ_RANGES]
    x = (rnd(B, 2, N) * 2 - 1).requires_grad_(True)
    w = torch.randn(B, 2, N, device=dev, generator=g)

    def step():
        x.grad = None
        for c in cols:
            c.grad = None
        D.parametric_eq(x, SR, *cols).backward(w)
    row = {"B": B, "rows": 2 * B, "N": N}
    for mode in ("0", "auto"):
        os.environ["DASP_SOS_SEGMENT"] = mode
        row["plain" if mode == "0" else "rule"] = round(graph_step_ms(step, replays=200, blocks=3, ramp_s=0.3), 4)
    print(json.dumps(row), flush=True)
