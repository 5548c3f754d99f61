"""Reverb fwd+bwd GPU-bound time at the reference's training shapes against the filter bank's band split (dasp_reverb_plan)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from bench import graph_step_ms
from dasp_pytorch_amd import _lib
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
out = {}
for B, C, N in ((16, 1, 131072), (12, 2, 131072), (20, 1, 131072), (24, 1, 131072)):
    x = (torch.rand(B, C, N, device=dev, generator=g) * 2 - 1).requires_grad_(True)
    ctl = [torch.rand(B, device=dev, generator=g).requires_grad_(True) for _ in range(25)]
    w = torch.randn(B, 2, N, device=dev, generator=g)
    def step():
        x.grad = None
        for c in ctl: c.grad = None
        D.noise_shaped_reverberation(x, 44100, *ctl, device_noise=True, noise_seed=5).backward(w)
    for split in (0, 1, 2, 3, 4):
        _lib.lib().dasp_reverb_plan(-1, -1.0, split if split else -1)
        out[f"({B},{C},{N}) split={split or 'plan'}"] = graph_step_ms(step, replays=60, blocks=3, ramp_s=0.2)
print(json.dumps(out, indent=1))
