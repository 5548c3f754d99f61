"""Kernels per eager forward + backward step of the reference's chain at its training shape (16, 1, 131072): run under
rocprofv3 --kernel-trace --stats and divide the call counts by STEPS (printed)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
STEPS = 50
g = torch.Generator(device="cuda").manual_seed(0)
chain = D.chain.StyleTransferChain(44100, device_noise=True)
xc = torch.rand(16, 1, 131072, device="cuda", generator=g) * 2 - 1
pcs = [(torch.rand(16, n, device="cuda", generator=g) * 0.9 + 0.05).requires_grad_(True) for n in chain.num_params]
wc = torch.randn(16, 2, 131072, device="cuda", generator=g)
torch.cuda.synchronize()
for _ in range(STEPS):
    for p in pcs: p.grad = None
    chain.process_normalized(xc, *pcs).backward(wc)
torch.cuda.synchronize()
print("STEPS", STEPS)
