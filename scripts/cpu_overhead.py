"""Developer check: host-side cost of one parametric_eq fwd+bwd step (tiny tensors => GPU time ~ 0)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
dev = "cuda:0"
B, C, N = 4, 2, 1024
x = torch.rand(B, C, N, device=dev).requires_grad_(True)
cols = [torch.rand(B, device=dev).add(1.0).mul(100).requires_grad_(True) for _ in range(18)]
w = torch.randn(B, C, N, device=dev)
def step():
    x.grad = None
    for c in cols: c.grad = None
    D.parametric_eq(x, 44100, *cols).backward(w)
for _ in range(200): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1000): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/1000:.3f} ms/step, with sync {1e3*(t2-t0)/1000:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
