#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json metric): audio channel-samples/sec of the 6-band parametric_eq
forward + backward at (B, C, N) = (256, 2, 131072) fp32 per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one call of dasp_pytorch_amd.functional.parametric_eq (coefficient design + cascade
forward) followed by the full backward (grad wrt x and all 18 controls) on one synthetic batch that
is already resident in HBM. Batches shard along the batch axis, one process per GPU, with no
data-path collective (the effect has no cross-item exchange): every rank processes its own
(256, 2, 131072) batch, so scaling is "weak" and `value` is the sum over ranks / max-over-ranks time.

The JSON line also carries
  roofline     -- HBM roofline of the dominant kernel (the backward cascade): algorithmic bytes per
                  launch / average launch duration measured with HIP events on the launch stream
  roofline_*   -- the same for the forward kernel and for forward+backward together
  cpu_baseline -- the numpy restatement of the reference's algorithm (oracle/dasp_oracle.py, kind
                  "port") timed on this box's host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import dasp_pytorch_amd as D  # noqa: E402
from dasp_pytorch_amd import _lib  # noqa: E402
from dasp_pytorch_amd import distributed as dd  # noqa: E402

SR = 44100
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# dasp_pytorch/modules.py:136-155 (ParametricEQ.param_ranges at sample_rate 44100), reference argument order
PEQ_RANGES = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
              (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]


def make_batch(B, C, N, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, C, N, generator=g) * 2 - 1
    pn = torch.rand(B, 18, generator=g)
    lo = torch.tensor([r[0] for r in PEQ_RANGES], dtype=torch.float32)
    hi = torch.tensor([r[1] for r in PEQ_RANGES], dtype=torch.float32)
    params = pn * (hi - lo) + lo
    w = torch.randn(B, C, N, generator=g)
    return x.to(device), params.to(device), w.to(device)


def cpu_baseline(seconds_budget=15.0):
    """Oracle (numpy port of the reference's frequency-sampling algorithm + its VJP) on the host."""
    from oracle import dasp_oracle as orc
    C, N = 2, 131072
    x, params, w = (t.numpy() for t in make_batch(4, C, N, 999, "cpu"))
    orc.parametric_eq(x[:1], SR, params[:1], dtype=np.float32)  # warm the FFT plans
    t0 = time.perf_counter()
    done = 0
    while done < 256 and time.perf_counter() - t0 < seconds_budget:  # 4 items at a time until the budget is spent
        orc.parametric_eq(x, SR, params, dtype=np.float32)
        orc.parametric_eq_vjp(x, SR, params, w, dtype=np.float32)
        done += 4
    dt = time.perf_counter() - t0
    return {"value": done * C * N / dt, "unit": "channel-samples/s", "cores": 1, "kind": "port",
            "sample": f"parametric_eq fwd+vjp fp32 on ({done},{C},{N}) of the (256,2,131072) workload, {dt:.1f} s, "
                      "numpy pocketfft single thread"}


def _time_steps(fn, steps=30, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def secondary(dev):
    """Short fwd+bwd timings of the other hot-path ops at their BASELINE.json configs (1 GPU, not the headline)."""
    res = {}
    g = torch.Generator(device=dev).manual_seed(7)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=g)

    def bench_op(name, B, C, N, make, bytes_per_cs, note=None, xmake=None):
        x = (xmake(B, C, N) if xmake else rnd(B, C, N) * 2 - 1).requires_grad_(True)
        ctl, call = make(B)
        w = torch.randn(B, 2 if name == "noise_shaped_reverberation" else C, N, device=dev, generator=g)

        def step():
            x.grad = None
            for c in ctl:
                c.grad = None
            call(x, ctl).backward(w)
        t = _time_steps(step)
        cs = B * C * N
        res[name] = {"shape": [B, C, N], "ms_fwd_bwd": round(t * 1e3, 3), "channel_samples_per_s": cs / t,
                     "algorithmic_GBps": round(bytes_per_cs * cs / t / 1e9, 1), "frac_of_8TBps": round(bytes_per_cs * cs / t / 1e9 / HBM_PEAK_GBS, 4)}
        if note:
            res[name]["note"] = note
        del x, w

    ctl1 = lambda lo, hi: (lambda B: (rnd(B) * (hi - lo) + lo).requires_grad_(True))
    bench_op("gain", 256, 2, 131072, lambda B: ([ctl1(-24, 24)(B)], lambda x, c: D.gain(x, SR, c[0])), 20)
    bench_op("distortion", 256, 2, 131072, lambda B: ([ctl1(0, 24)(B * 2)], lambda x, c: D.distortion(x, SR, c[0])), 20)
    rng = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]
    bench_op("compressor", 256, 2, 262144, lambda B: ([ctl1(lo, hi)(B) for lo, hi in rng], lambda x, c: D.compressor(x, SR, *c)), 20)


    def speechlike(B, C, N):   # white noise x a slow random envelope spanning -60 .. 0 dBFS (all three knee regions)
        knots = rnd(B, 1, N // 4096 + 2) * -60.0
        env_db = torch.nn.functional.interpolate(knots, size=N, mode="linear", align_corners=True)
        return (rnd(B, C, N) * 2 - 1) * torch.pow(10.0, env_db / 20.0)
    bench_op("compressor_speechlike", 256, 2, 262144, lambda B: ([ctl1(lo, hi)(B) for lo, hi in rng], lambda x, c: D.compressor(x, SR, *c)), 20,
             "same op on white noise x slow random envelope, -60 .. 0 dBFS", xmake=speechlike)
    bench_op("noise_shaped_reverberation", 128, 2, 262144,
             lambda B: ([ctl1(0, 1)(B) for _ in range(25)], lambda x, c: D.noise_shaped_reverberation(x, SR, *c, device_noise=True)),
             2 * 1.354e9 / (128 * 2 * 262144),
             "device-generated noise; bytes = SURVEY 8(d) compulsory traffic with noise as an input (2 x 1.354 GB)")
    # widening rows: stereo utilities and the multi-resolution STFT loss
    bench_op("stereo_widener", 256, 2, 131072, lambda B: ([ctl1(0, 1)(B)], lambda x, c: D.stereo_widener(x, SR, c[0].reshape(-1, 1))), 20)
    xs = (rnd(16, 2, 131072) * 0.6 - 0.3).requires_grad_(True)
    ys = rnd(16, 2, 131072) * 0.6 - 0.3
    loss_fn = D.losses.MultiResolutionSTFTLoss()

    def loss_step():
        xs.grad = None
        loss_fn(xs, ys).backward()
    t = _time_steps(loss_step)
    res["mrstft_loss"] = {"shape": [16, 2, 131072], "ms_fwd_bwd": round(t * 1e3, 3), "channel_samples_per_s": 16 * 2 * 131072 / t,
                          "note": "3 resolutions (1024/120/600, 2048/240/1200, 512/50/240); compute-bound (7.4 transforms per input sample "
                                  "and direction), HBM traffic is the two signals and the gradient"}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--ramp-seconds", type=float, default=1.0,
                    help="untimed clock ramp before the warmup steps: the MI355X needs ~0.2 s of sustained load to leave its idle "
                         "clocks (measured: the same kernels run 1.28x slower in the first 10 ms)")
    ap.add_argument("--batch", type=int, default=256, help="batch items per GPU")
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--samples", type=int, default=131072)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short timings of the other hot-path ops")
    ap.add_argument("--no-kernel-events", action="store_true", help="developer switch: do not record per-kernel HIP events in the timed region")
    args = ap.parse_args()

    rank, local, world = dd.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists in dasp_pytorch_amd)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dd.init("nccl", dev)

    B, C, N = args.batch, args.channels, args.samples
    x, params, w = make_batch(B, C, N, 1234 + rank, dev)
    x.requires_grad_(True)
    cols = [params[:, i].clone().requires_grad_(True) for i in range(18)]

    def step():
        x.grad = None
        for c in cols:
            c.grad = None
        y = D.parametric_eq(x, SR, *cols)
        y.backward(w)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # device clock ramp (untimed, before the W warmup steps; reported as "ramp_s")
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_kernel_events:
        _lib.timers.start(every=4)   # HIP events around every 4th launch of each entry point, inside the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ktimes = _lib.timers.stop() if not args.no_kernel_events else {"dasp_sosfilt_forward": [float("nan")], "dasp_sosfilt_backward_ex": [float("nan")]}
    dt = dd.max_over_ranks(dt, dev)
    finite = bool(torch.isfinite(x.grad).all().item()) and all(bool(torch.isfinite(c.grad).all().item()) for c in cols)

    if rank == 0:
        units = B * C * N                      # channel-samples per step per GPU
        ms = dt / args.steps * 1e3
        value = units * world / (dt / args.steps)
        t_fwd = float(np.mean(ktimes["dasp_sosfilt_forward"])) * 1e-3
        t_bwd = float(np.mean(ktimes["dasp_sosfilt_backward_ex"])) * 1e-3
        t_small = sum(float(np.mean(v)) for k, v in ktimes.items() if k not in ("dasp_sosfilt_forward", "dasp_sosfilt_backward_ex")) * 1e-3

        # HBM traffic per launch from the PMC counters: collected off-line (rocprofv3 --pmc passes cannot run inside
        # this process) on the same shape and stored under profiles/; null when the shape differs
        traffic = {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01", "hbm_traffic.json")))
            if tj["shape"] == [B, C, N]:
                traffic = {"fwd": tj["sos_fwd_kernel"]["hbm_bytes"], "bwd": tj["sos_bwd_kernel"]["hbm_bytes"]}
                traffic["both"] = traffic["fwd"] + traffic["bwd"]
        except (OSError, KeyError, ValueError):
            pass

        def roof(bytes_per_sample, t, which=None):
            a = bytes_per_sample * units / t / 1e9
            return {"bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4),
                    "traffic": traffic.get(which)}

        out = {
            "metric": "audio-samples/sec fwd+bwd, 6-band parametric_eq @ (256,2,131072)",
            "value": value, "unit": "channel-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "ramp_s": args.ramp_seconds,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"parametric_eq fwd+bwd (grad x + 18 controls) on ({B},{C},{N}) fp32 per GPU, sr 44100, "
                                   "controls ~ U(ParametricEQ ranges)", "global_batch": B * world,
                       "parallelism": f"batch-shard x{world}, no collective"},
            "roofline": dict(roof(12, t_bwd, "bwd"), kernel="sos_bwd_kernel<6>", ms=round(t_bwd * 1e3, 4),
                             algorithmic_bytes=12 * units),
            "roofline_fwd": dict(roof(8, t_fwd, "fwd"), kernel="sos_fwd_kernel<6>", ms=round(t_fwd * 1e3, 4), algorithmic_bytes=8 * units),
            "roofline_fwd_bwd": dict(roof(20, t_fwd + t_bwd, "both"), ms=round((t_fwd + t_bwd) * 1e3, 4), algorithmic_bytes=20 * units),
            "small_kernels_ms": round(t_small * 1e3, 4),
            "finite": finite,
        }
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
