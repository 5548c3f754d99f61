"""Developer tool: phase time stamps of one wave of dyn_bwd_kernel (library built with -DDASP_TRACE, see tools/README.md)."""
import ctypes, os, sys
os.environ["DASP_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "_dbg", "libdasp_trace.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dasp_pytorch_amd as D
from dasp_pytorch_amd import _lib
B, C, N = 256, 2, 262144
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(B, C, N, device=dev, generator=g) * 2 - 1).requires_grad_(True)
rng = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]
ctl = [(torch.rand(B, device=dev, generator=g) * (hi - lo) + lo).requires_grad_(True) for lo, hi in rng]
w = torch.randn(B, C, N, device=dev, generator=g)
for _ in range(200):
    x.grad = None
    D.compressor(x, 44100, *ctl).backward(w)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ["DASP_HIP_LIB"])
out = (ctypes.c_longlong * 64)()
assert L.dasp_debug_dyn_trace(out) == 0
names = ["start", "loads+sum", "fwd recompute", "fwd g / q", "adj scan", "mbox wait", "publish", "adjoint", "stores"]
for k in range(4):
    v = [out[k * 8 + i] for i in range(8)]
    print("tile", k, " ".join(f"{names[i + 1]}={v[i + 1] - v[i]}" for i in range(7)), "total", v[7] - v[0])
