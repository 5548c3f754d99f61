"""Developer timing (round 5): does cutting rows into segments pay beyond 64 rows now that a segmented step is three launches?
EQ fwd+bwd as a replayed HIP graph at (B, 2, 131072): one workgroup per row (twice the waves up to 256 rows) against forced segment lengths."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from bench import graph_step_ms, PEQ_RANGES, SR
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
N = 131072
for B in (24, 32, 40, 48, 64, 96, 128):
    cols = [(rnd(B) * (hi - lo) + lo).requires_grad_(True) for lo, hi in PEQ_RANGES]
    x = (rnd(B, 2, N) * 2 - 1).requires_grad_(True)
    w = torch.randn(B, 2, N, device=dev, generator=g)
    def step():
        x.grad = None
        for c in cols: c.grad = None
        D.parametric_eq(x, SR, *cols).backward(w)
    row = {"B": B, "rows": 2 * B}
    os.environ["DASP_SOS_SEGMENT"] = "0"; os.environ.pop("DASP_SOS_SEGMENT_TILES", None)
    row["plain"] = round(graph_step_ms(step, replays=100, blocks=3, ramp_s=0.2), 4)
    os.environ["DASP_SOS_SEGMENT"] = "auto"
    row["rule"] = round(graph_step_ms(step, replays=100, blocks=3, ramp_s=0.2), 4)
    os.environ["DASP_SOS_SEGMENT"] = "1"
    for T in (8, 16, 32, 64):
        os.environ["DASP_SOS_SEGMENT_TILES"] = str(T)
        row[f"T={T}"] = round(graph_step_ms(step, replays=100, blocks=3, ramp_s=0.2), 4)
    os.environ.pop("DASP_SOS_SEGMENT_TILES", None)
    print(json.dumps(row), flush=True)
