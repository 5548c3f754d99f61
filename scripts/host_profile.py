"""Developer tool: where the eager host time of the reference's training-shape chain goes (cProfile of the Python side of
StyleTransferChain fwd + bwd on (16, 1, 131072) with normalised parameters; GPU work is asynchronous, the profile is host time only)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd.chain import StyleTransferChain
g = torch.Generator(device="cuda:0").manual_seed(0)
chain = StyleTransferChain(44100, device_noise=True)
x = torch.rand(16, 1, 131072, device="cuda:0", generator=g) * 0.6 - 0.3
ps = [(torch.rand(16, n, device="cuda:0", generator=g) * 0.9 + 0.05).requires_grad_(True) for n in chain.num_params]
w = torch.randn(16, 2, 131072, device="cuda:0", generator=g)
def step():
    for p in ps: p.grad = None
    chain.process_normalized(x, *ps).backward(w)
for _ in range(30): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("wall per step %.3f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
torch.autograd.set_multithreading_enabled(False)          # backward in this thread, so that the profile sees the Python backward functions
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(34)
