"""Developer GPU check (not a test): parity on golden cases + a quick timing. Run via gpurun."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D
from dasp_pytorch_amd import signal as DS
from oracle import dasp_oracle as orc
from tests.util import load_golden, linf_peak
SR = 44100
dev = "cuda:0"
print(torch.cuda.get_device_name(0))

def t(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

for name in ["eq_b3c2_n12000", "eq_bcast_b2c1_n4099"]:
    g = load_golden(name)
    x = t(g["x"]).requires_grad_(True)
    cols = [t(g["params"][:, i]).requires_grad_(True) for i in range(18)]
    y = D.parametric_eq(x, SR, *cols)
    (y * t(g["w"])).sum().backward()
    torch.cuda.synchronize()
    gp = torch.stack([c.grad for c in cols], 1).cpu().numpy()
    print(name, "y vs ref64", linf_peak(y.detach().cpu().numpy(), g["y64"]), " ref32 vs ref64", linf_peak(g["y32"], g["y64"]))
    print("   gx vs ref64", linf_peak(x.grad.cpu().numpy(), g["gx64"]), " ref32:", linf_peak(g["gx32"], g["gx64"]))
    print("   gp vs ref64", linf_peak(gp, g["gp64"]), " ref32:", linf_peak(g["gp32"], g["gp64"]))

g = load_golden("sos_b2c2_n6000_s3")
x = t(g["x"]).requires_grad_(True); sos = t(g["sos"]).requires_grad_(True)
y = DS.sosfilt_via_fsm(sos, x); (y * t(g["w"])).sum().backward(); torch.cuda.synchronize()
print("sos y", linf_peak(y.detach().cpu().numpy(), g["y64"]), "gx", linf_peak(x.grad.cpu().numpy(), g["gx64"]),
      "gsos", linf_peak(sos.grad.cpu().numpy(), g["gsos64"]), " ref32 gsos:", linf_peak(g["gsos32"], g["gsos64"]))

# medium case vs numpy oracle (fp64), full north-star length
gen = torch.Generator().manual_seed(7)
B, C, N = 6, 2, 131072
xn = (torch.rand(B, C, N, generator=gen) * 2 - 1)
g2 = load_golden("eq_b3c2_n12000")
pn = torch.from_numpy(np.concatenate([g2["params"], g2["params"][::-1]], 0).copy())
wn = torch.randn(B, C, N, generator=gen)
x = xn.to(dev).requires_grad_(True); cols = [pn[:, i].to(dev).requires_grad_(True) for i in range(18)]
y = D.parametric_eq(x, SR, *cols); (y * wn.to(dev)).sum().backward(); torch.cuda.synchronize()
yo = orc.parametric_eq(xn.numpy(), SR, pn.numpy()); gxo, gpo = orc.parametric_eq_vjp(xn.numpy(), SR, pn.numpy(), wn.numpy())
print("N=131072 y", linf_peak(y.detach().cpu().numpy(), yo)); print("  gx", linf_peak(x.grad.cpu().numpy(), gxo))
print("  gp", linf_peak(torch.stack([c.grad for c in cols], 1).cpu().numpy(), gpo))

# timing at the north-star shape
B, C, N = 256, 2, 131072
x = (torch.rand(B, C, N, device=dev) * 2 - 1).requires_grad_(True)
mod_ranges = [(-20,20),(20,2000),(.1,6),(-20,20),(80,2000),(.1,6),(-20,20),(2000,8000),(.1,6),(-20,20),(8000,12000),(.1,6),(-20,20),(12000,21050),(.1,6),(-20,20),(4000,21050),(.1,6)]
cols = [(torch.rand(B, device=dev) * (hi - lo) + lo).requires_grad_(True) for lo, hi in mod_ranges]
w = torch.randn(B, C, N, device=dev)
for it in range(3):
    y = D.parametric_eq(x, SR, *cols); y.backward(w)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
K = 10; tf = tb = 0
for it in range(K):
    e0.record(); y = D.parametric_eq(x, SR, *cols); e1.record(); y.backward(w); e2.record(); torch.cuda.synchronize()
    tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
print(f"fwd {tf/K:.3f} ms  bwd {tb/K:.3f} ms  total {(tf+tb)/K:.3f} ms -> {B*C*N/((tf+tb)/K*1e-3):.3e} samples/s, "
      f"{20*B*C*N/((tf+tb)/K*1e-3)/1e12:.2f} TB/s algorithmic")
print("finite:", torch.isfinite(y).all().item(), torch.isfinite(x.grad).all().item())
