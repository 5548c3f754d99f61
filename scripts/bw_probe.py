"""Read-only / write-only / copy bandwidth of the box with plain torch kernels (context for the roofline fractions in DESIGN.md)."""
import torch, time
def t(f, n=50):
    for _ in range(10): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for mb in (256, 1024, 4096):
    n = mb*1024*1024//4
    a = torch.empty(n, device='cuda'); b = torch.empty(n, device='cuda'); a.normal_()
    w = t(lambda: b.fill_(1.0)); c = t(lambda: b.copy_(a)); r = t(lambda: a.sum())
    add = t(lambda: torch.add(a, a, out=b))
    print(f"{mb} MB: fill {mb/1024/w*1e3/1.024:.0f} GB/s  copy(r+w) {2*mb/1024/c*1e3/1.024:.0f} GB/s  sum(read) {mb/1024/r*1e3/1.024:.0f} GB/s  add(r+w) {2*mb/1024/add*1e3/1.024:.0f} GB/s", flush=True)
