#!/bin/bash
# Developer helper: shader clock and package power (rocm-smi, sampled every 0.5 s) while a workload loops. usage: scripts/clock_probe.sh eq|reverb|idle
what=${1:-eq}
python - "$what" <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
import dasp_pytorch_amd as D
what = sys.argv[1]
g = torch.Generator(device="cuda").manual_seed(0)
if what == "eq":
    B, C, N = 256, 2, 131072
    x = (torch.rand(B, C, N, device="cuda", generator=g) * 2 - 1).requires_grad_(True)
    pn = torch.rand(B, 18, device="cuda", generator=g).requires_grad_(True)
    w = torch.randn(B, C, N, device="cuda", generator=g)
    eq = D.ParametricEQ(44100)
    def step(): eq.process_normalized(x, pn).backward(w); x.grad = None; pn.grad = None
elif what == "reverb":
    B, C, N = 128, 2, 262144
    x = (torch.rand(B, C, N, device="cuda", generator=g) * 2 - 1).requires_grad_(True)
    ctl = [torch.rand(B, device="cuda", generator=g).requires_grad_(True) for _ in range(25)]
    w = torch.randn(B, 2, N, device="cuda", generator=g)
    def step(): D.noise_shaped_reverberation(x, 44100, *ctl, device_noise=True, noise_seed=1).backward(w); x.grad = None
else:
    def step(): time.sleep(0.01)
t_end = time.perf_counter() + 8.0
n = 0
while time.perf_counter() < t_end:
    step(); n += 1
torch.cuda.synchronize()
print(what, "steps", n, "ms/step", 8000.0 / n)
PY
pid=$!
sleep 3.5
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo; sleep 0.5; done
wait $pid
