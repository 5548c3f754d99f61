"""lfilter_via_fsm (K > 3) forward + backward as one HIP graph replayed back to back: GPU-bound time (bench.graph_step_ms), per chunk length.
usage: python scripts/lfilter_graph_time.py"""
import os, sys, json
import numpy as np, scipy.signal, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from bench import graph_step_ms
out = {}
for B, N, K in ((16, 262144, 5), (256, 262144, 5), (16, 262144, 16)):
    ba = [scipy.signal.butter(K - 1, 0.3) for _ in range(B)]
    b = torch.tensor(np.stack([q[0] for q in ba]), dtype=torch.float32, device="cuda").requires_grad_(True)
    a = torch.tensor(np.stack([q[1] for q in ba]), dtype=torch.float32, device="cuda").requires_grad_(True)
    x = (torch.rand(B, 1, N, device="cuda") * 2 - 1).requires_grad_(True)
    w = torch.randn(B, 1, N, device="cuda")
    def step():
        x.grad = None; b.grad = None; a.grad = None
        D.signal.lfilter_via_fsm(x, b, a).backward(w)
    for chunk in (0, 64, 128, 256, 512, 1024, 2048):
        os.environ["DASP_LFILTER_CHUNK"] = str(chunk)
        out[f"({B},1,{N}) K={K} chunk={chunk or 'default'}"] = graph_step_ms(step, replays=10, blocks=3, ramp_s=0.1)
print(json.dumps(out, indent=1))
