"""Wall time per training-like step of the five-effect chain at the reference's batch (16, 1, 131072): issued from Python every step
against torch.cuda.make_graphed_callables over StyleTransferChain.process_normalized (forward and backward as HIP-graph replays inside an
otherwise ordinary autograd step). usage: python scripts/chain_graphed_time.py [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
SR = 44100
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = 131072
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
off = torch.zeros(1, dtype=torch.int64, device=dev)
chain = D.chain.StyleTransferChain(SR, device_noise=True, noise_seed=7, noise_seed_offset=off)
x = torch.rand(B, 1, N, device=dev, generator=g) * 2 - 1
ps = [(torch.rand(B, n, device=dev, generator=g) * 0.9 + 0.05).requires_grad_(True) for n in chain.num_params]
w = torch.randn(B, 2, N, device=dev, generator=g)
a = torch.rand(1 << 26, device=dev); t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end: a.mul_(1.0001)
torch.cuda.synchronize()
fn = lambda x_, a_, b_, c_, d_: chain.process_normalized(x_, a_, b_, c_, d_)
graphed = torch.cuda.make_graphed_callables(fn, (x.clone(),) + tuple(p.detach().clone().requires_grad_(True) for p in ps))


def step(f):
    for p in ps: p.grad = None
    off.add_(1)
    f(x, *ps).backward(w)


def wall(f, n=200):
    for _ in range(20): step(f)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n): step(f)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[2]


te = wall(fn); tg = wall(graphed); te2 = wall(fn)
print(f"chain step ({B},1,{N}) fwd+bwd wall: eager {te:.3f} ms (again {te2:.3f}), make_graphed_callables {tg:.3f} ms  -> {te / tg:.2f}x")
