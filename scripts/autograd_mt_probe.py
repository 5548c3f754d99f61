import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dasp_pytorch_amd as D
SR=44100
PEQ = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
       (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
g = torch.Generator(device="cuda").manual_seed(0)
for B in (16, 256):
    x = (torch.rand(B, 2, 131072, device="cuda", generator=g) * 2 - 1).requires_grad_(True)
    w = torch.randn(B, 2, 131072, device="cuda", generator=g)
    cols = [(torch.rand(B, device="cuda", generator=g) * (hi - lo) + lo).requires_grad_(True) for lo, hi in PEQ]
    def step():
        x.grad = None
        for c in cols: c.grad = None
        D.parametric_eq(x, SR, *cols).backward(w)
    for mt in (True, False, True, False):
        torch.autograd.set_multithreading_enabled(mt)
        for _ in range(100): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): step()
        torch.cuda.synchronize()
        print(f"B={B} autograd multithreading={mt}: wall {(time.perf_counter()-t0)/300*1e3:.4f} ms per step")
