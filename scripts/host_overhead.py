"""Developer timing: eager wall time and GPU time per forward + backward step of the normalised-parameter API at a training-size batch,
for the generic path (18 / 6 / 25 column views through the functional signatures) and the fused / matrix paths of modules.py."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import _lib, modules
SR = 44100
B, C, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 2, 131072)))
g = torch.Generator(device="cuda:0").manual_seed(0)
x = torch.rand(B, C, N, device="cuda:0", generator=g) * 2 - 1
w = torch.randn(B, 2, N, device="cuda:0", generator=g)


def timeit(fn, n=200):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / n
    _lib.timers.start(every=1)
    for _ in range(20): fn()
    gpu = sum(sum(v) for v in _lib.timers.stop().values()) / 20
    return wall * 1e3, gpu


for name, mod in (("ParametricEQ", D.ParametricEQ(SR)), ("Compressor", D.Compressor(SR)), ("NoiseShapedReverb", D.NoiseShapedReverb(SR, device_noise=True))):
    p = (torch.rand(B, mod.num_params, device="cuda:0", generator=g) * 0.9 + 0.05).requires_grad_(True)
    def step(generic):
        p.grad = None
        y = modules.Processor.process_normalized(mod, x, p) if generic else mod.process_normalized(x, p)
        y.backward(w[:, :y.shape[1]])
    wg, gg = timeit(lambda: step(True))
    wf, gf = timeit(lambda: step(False))
    print(f"{name} ({B},{C},{N}) process_normalized fwd+bwd: generic path wall {wg:.3f} ms (GPU {gg:.3f}), fused / matrix path wall {wf:.3f} ms (GPU {gf:.3f})", flush=True)
