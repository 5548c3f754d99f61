import sys, time, torch, numpy as np, scipy.signal
sys.path.insert(0, ".")
import dasp_pytorch_amd as D
for B, N, K in ((16, 262144, 5), (256, 262144, 5), (16, 262144, 16), (16, 131072, 8)):
    ba = [scipy.signal.butter(K - 1, 0.3) for _ in range(B)]
    b = torch.tensor(np.stack([q[0] for q in ba]), dtype=torch.float32, device="cuda").requires_grad_(True)
    a = torch.tensor(np.stack([q[1] for q in ba]), dtype=torch.float32, device="cuda").requires_grad_(True)
    x = (torch.rand(B, 1, N, device="cuda") * 2 - 1).requires_grad_(True)
    w = torch.randn(B, 1, N, device="cuda")
    def step():
        x.grad = None; b.grad = None; a.grad = None
        D.signal.lfilter_via_fsm(x, b, a).backward(w)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize()
    print(f"lfilter_via_fsm ({B},1,{N}) K={K} fwd+bwd {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
