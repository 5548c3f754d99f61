#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json metric): audio channel-samples/sec of the 6-band parametric_eq
forward + backward at (B, C, N) = (256, 2, 131072) fp32.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--launch eager|graph]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one call of dasp_pytorch_amd.functional.parametric_eq (coefficient design + cascade
forward) followed by the full backward (grad wrt x and all 18 controls) on one synthetic batch that
is already resident in HBM. Batches shard along the batch axis, one process per GPU, with no
data-path collective (the effect has no cross-item exchange).
  --scaling weak   (default) every rank processes its own (256, 2, 131072) batch; `value` = N * units / max-over-ranks time
  --scaling strong the one 256-item batch is partitioned 256/N items per GPU; `value` = units / max-over-ranks time
  --launch graph   the step (forward + backward, same kernels, same launches) is captured once into a HIP graph and replayed;
                   eager issues it from Python every step. --launch both (default) times both: `launch_ms_per_step` has both,
                   `value` / `ms_per_step` are the faster one and `config.launch` names it (the step is 0.40 ms of kernels behind ~0.3 ms of
                   host work per eager step, so a slow or loaded host makes the eager figure a host measurement).
  --gpus N         N ranks. Under torchrun (WORLD_SIZE set) the world size must equal N; without it and N > 1, bench.py re-launches
                   itself as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`. `n_gpus` = ranks that joined the
                   process group.

Timing. Each launch mode is measured on the product path as a user calls it (one C call per direction, no timers, no events): an
untimed clock ramp issued in that mode (--ramp-seconds), W warm-up steps, then --blocks blocks of EXACTLY K steps, each bracketed by
barrier + device synchronize on both sides, max over ranks; `value` / `ms_per_step` are the median block (`block_ms_per_step` has
min / median / max). A 20-step block is 8 ms, shorter than the 0.2 s the shader clock needs to ramp and of the same order as a host
hiccup: the ramp precedes every mode and the median keeps a single disturbed block out of `value`.

The JSON line also carries
  roofline     -- HBM roofline of the dominant kernel (the backward cascade): algorithmic bytes per launch / average launch duration,
                  measured live with HIP events on the launch stream in a separate, untimed pass directly behind the timed blocks (the
                  events need the kernels as separate entry points, which is not how the product issues them, so they stay out of the
                  timed region; 240 back-to-back steps, events around every 4th launch of each entry point, queued behind a backlog of
                  graph replays sized from a probe of the host's issue rate so that the GPU never waits for the host - a waiting GPU
                  runs these power-limited kernels faster than the timed step does; --no-event-backlog turns the backlog off);
                  `kernel_events_over_step` = the four kernels' event durations over the timed step (0.97 - 1.03 when consistent);
                  `traffic` = HBM bytes per launch from the PMC counters, read from profiles/<round>/hbm_traffic.json only if that
                  file was produced from the kernel sources now loaded
  frames_per_s, step_algorithmic_GBps_per_gpu -- the same rate in frames (B N) and the whole step on its 20 B per channel-sample
  roofline_*   -- the same for the forward kernel and for forward+backward together
  cpu_baseline -- the reference itself (oracle/_ref, staged by __graft_entry__.build(); kind "reference") on the host cores on a
                  bounded sub-batch (its best-case batch size), or, when it is not staged, the numpy restatement
                  (oracle/dasp_oracle.py, kind "port").

--dry-run-cpu is a test hook (tests/test_distributed_cpu.py): rank wiring, sharding, barriers, timing reduction and the JSON contract
on CPU tensors over gloo with the kernels replaced by a copy; its numbers mean nothing and the line says so.
"""
import argparse
import glob
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
REF_ZIP = os.path.join(ROOT, "oracle", "_ref", "dasp_pytorch_ref.zip")


def make_batch(B, C, N, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, C, N, generator=g) * 2 - 1
    pn = torch.rand(B, 18, generator=g)
    lo = torch.tensor([r[0] for r in PEQ_RANGES], dtype=torch.float32)
    hi = torch.tensor([r[1] for r in PEQ_RANGES], dtype=torch.float32)
    params = pn * (hi - lo) + lo
    w = torch.randn(B, C, N, generator=g)
    return x.to(device), params.to(device), w.to(device)


def kernel_source_hash():
    """Code hash (comments and whitespace ignored) of the cascaded-biquad kernel sources the loaded library was built from: ties the
    off-line counter file to the kernels it measured (dasp_pytorch_amd/csrc/build.py)."""
    from dasp_pytorch_amd.csrc.build import kernel_source_hash as h
    return h()


def cpu_baseline_reference(seconds_budget=25.0, full=False):
    """The reference's own dasp_pytorch.functional.parametric_eq (imported from the archive oracle/_ref holds) forward + autograd backward
    on all host cores, fp32, on a bounded sub-batch of the workload - (8,2,131072), the reference's best case: its throughput FALLS with
    the batch. full=True (--cpu-baseline-full): the whole (256,2,131072) workload instead, once per thread count tried -
    minutes of host time and ~14 GB of host memory, so not part of the default run (profiles/r06/cpu_baseline_full.json)."""
    if REF_ZIP not in sys.path:
        sys.path.insert(0, REF_ZIP)
    import dasp_pytorch.functional as RF
    cores = os.cpu_count() or 1
    B, C, N = (256, 2, 131072) if full else (8, 2, 131072)
    x, params, w = make_batch(B, C, N, 999, "cpu")

    def step():
        xx = x.clone().requires_grad_(True)
        cols = [params[:, i].clone().requires_grad_(True) for i in range(18)]
        y = RF.parametric_eq(xx, SR, *cols)
        y.backward(w)
    # torch's CPU FFTs and elementwise ops stop scaling well before a 256-thread host is used up (measured on a gpurun box: 0.21 s per
    # iteration with 16 threads, 0.46 s with 64, 18.4 s with all 256): thread counts are tried in increasing order until the time budget
    # is spent or more threads made it slower, and the best one is reported
    tried = {}
    t_all = time.perf_counter()
    for threads in (sorted({min(cores, 16), min(cores, 64)}) if full else sorted({min(cores, 16), min(cores, 64), cores})):
        if tried and (time.perf_counter() - t_all > seconds_budget or (len(tried) > 1 and list(tried.values())[-1] > list(tried.values())[-2])):
            break                            # out of budget, or more threads already made it slower (256 threads: 18 s per iteration)
        torch.set_num_threads(threads)
        if not full:
            step()                           # warm-up (FFT plans, allocator, thread pool)
        times = []
        for _ in range(1 if full else 2):
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
        tried[threads] = min(times)
    threads = min(tried, key=tried.get)
    best = tried[threads]
    return {"value": B * C * N / best, "unit": "channel-samples/s", "cores": threads, "kind": "reference",
            "sample": f"dasp_pytorch.functional.parametric_eq fwd + autograd bwd, fp32, on ({B},{C},{N}) of the (256,2,131072) workload, "
                      + ("one cold iteration per thread count (the full workload: --cpu-baseline-full)" if full else "best of 2 after 1 warm-up per thread count")
                      + ", s per iteration by threads: "
                      + ", ".join(f"{k}: {v:.2f}" for k, v in tried.items()) + f" (host has {cores}); torch {torch.__version__} CPU"}


def cpu_baseline_port(seconds_budget=15.0):
    """Fallback when the reference is not staged: the oracle (numpy port of the reference's frequency-sampling algorithm + its VJP)."""
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
                      "numpy pocketfft single thread (oracle/_ref not staged: run __graft_entry__.build() where /root/reference exists)"}


def cpu_baseline(full=False):
    if os.path.exists(REF_ZIP):
        try:
            return cpu_baseline_reference(full=full)
        except Exception as e:      # a broken archive must not cost the bench line; say what happened
            out = cpu_baseline_port()
            out["sample"] += f" [reference baseline failed: {type(e).__name__}: {e}]"
            return out
    return cpu_baseline_port()


def _time_steps(fn, steps=30, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _time_steps_median(fn, steps=30, warmup=10, blocks=3):
    """Eager wall time per step, median of `blocks` blocks (one block of 30 steps read 0.67 ms where its neighbours read 0.39: host jitter)."""
    ts = [_time_steps(fn, steps=steps, warmup=warmup if i == 0 else 2) for i in range(blocks)]
    return float(np.median(ts))


def graph_step_ms(step, replays=50, blocks=5, ramp_s=0.3):
    """GPU-bound milliseconds per step, comparable across boxes: `step` (forward + backward into static .grad buffers) is captured once into
    a HIP graph and replayed back to back - the host issues a replay in ~15 us, so the queue never runs dry, and there is no event pair
    between kernels (events around the library calls of an eager, host-bound loop insert idle gaps after which power-limited kernels clock
    up: round 3's `gpu_ms` of the small-batch ops differed by 36 % between boxes). Median of `blocks` blocks of `replays` replays after a
    sustained ramp of the same replays."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                 # warm-up off the default stream: allocator pools, cached tables, lazy module state
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp_s:
        for _ in range(25):
            graph.replay()
        torch.cuda.synchronize()
    ts = []
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(replays):
            graph.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / replays * 1e3)
    del graph
    return round(float(np.median(ts)), 4)


def secondary_traffic(name):
    """HBM bytes per fwd+bwd step of a secondary op by the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes), from the newest
    counter file under profiles/ that covers it: the reverb's hbm_traffic_secondary.json (scripts/reverb_traffic.sh), the compressor /
    expander / gain / distortion's hbm_traffic_ops.json (scripts/ops_traffic.sh; used only while its source hash equals the hash of the
    kernel sources of this build). null when no counter file covers the op."""
    if name == "noise_shaped_reverberation":
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic_secondary.json")), reverse=True):
            try:
                tj = json.load(open(path))
                if tj.get("noise_mode", "explicit") == "generated" and "hbm_bytes_per_step" in tj:
                    return {"bytes": int(tj["hbm_bytes_per_step"]), "file": os.path.relpath(path, ROOT)}
            except (OSError, ValueError):
                continue
        return None
    family = {"compressor": "dynamics", "expander": "dynamics", "gain": "elementwise", "distortion": "elementwise"}.get(name)
    if family is None:
        return None
    from dasp_pytorch_amd.csrc.build import kernel_source_hash as h
    now = h(("dynamics.hip", "dyn_common.hpp", "common.hpp")) if family == "dynamics" else h(("elementwise.hip", "common.hpp"))
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic_ops.json")), reverse=True):
        try:
            tj = json.load(open(path))
            op = tj["ops"][name]
            if tj["kernel_source_hash"][family] == now and "hbm_bytes_per_step" in op:
                return {"bytes": int(op["hbm_bytes_per_step"]), "ratio_to_algorithmic": op["ratio"], "file": os.path.relpath(path, ROOT)}
        except (OSError, ValueError, KeyError):
            continue
    return None


def mrstft_roofline(gpu_ms):
    """The MR-STFT loss is compute-bound: its roofline is the fp32 vector peak. VALU instruction counts per launch come from the counter
    file of scripts/mrstft_roofline.sh (profiles/rNN/mrstft_roofline.json, (16, 2, 131072), default resolutions); the duration is this
    run's. frac = VALU instructions x 64 lanes / time / (256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz)."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "mrstft_roofline.json")), reverse=True):
        try:
            tj = json.load(open(path))
            valu = sum(k["valu_insts"] for n, k in tj["kernels"].items() if "split_kernel" in n)
            hbm = sum(k.get("hbm_bytes", 0) for n, k in tj["kernels"].items() if "split_kernel" in n)
            peak = 256 * 4 * 32 * 2.4e9
            return {"bound": "valu", "achieved": round(valu * 64 / (gpu_ms * 1e-3) / 1e12, 2), "peak": round(peak / 1e12, 1), "unit": "T lane-op/s",
                    "frac": round(valu * 64 / (gpu_ms * 1e-3) / peak, 4), "valu_instructions_per_step": int(valu),
                    "hbm_bytes_per_step": int(hbm), "hbm_frac_of_8TBps": round(hbm / (gpu_ms * 1e-3) / 8e12, 4), "counter_file": os.path.relpath(path, ROOT)}
        except (OSError, ValueError, KeyError):
            continue
    return None


def secondary(dev, skip_default_noise=False):
    """Short fwd+bwd timings of the other hot-path ops at their BASELINE.json configs (1 GPU, not the headline)."""
    res = {}
    g = torch.Generator(device=dev).manual_seed(7)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=g)

    def bench_op(name, B, C, N, make, bytes_per_cs, note=None, xmake=None, x_grad=True):
        x = (xmake(B, C, N) if xmake else rnd(B, C, N) * 2 - 1).requires_grad_(x_grad)
        ctl, call = make(B)
        w = torch.randn(B, 2 if name.startswith("noise_shaped_reverberation") else C, N, device=dev, generator=g)

        def step():
            x.grad = None
            for c in ctl:
                c.grad = None
            call(x, ctl).backward(w)
        t = _time_steps_median(step)                 # eager wall per step, median of three blocks of 30 steps
        cs = B * C * N
        res[name] = {"shape": [B, C, N], "ms_fwd_bwd": round(t * 1e3, 3), "channel_samples_per_s": cs / t,
                     "algorithmic_GBps": round(bytes_per_cs * cs / t / 1e9, 1), "frac_of_8TBps": round(bytes_per_cs * cs / t / 1e9 / HBM_PEAK_GBS, 4)}
        # GPU time of the step's library calls (HIP events around every C entry point), median of three blocks of ten steps: a cross-check
        # of the figure the roofline uses (below) - event pairs in an eager loop leave idle gaps after which power-limited kernels clock
        # differently, so this one wanders by box and by run (r05: 0.536 where the wall said 0.483)
        blocks_ms = []
        for _ in range(3):
            _lib.timers.start(every=1)
            for _ in range(10):
                step()
            kt = _lib.timers.stop()
            blocks_ms.append(sum(sum(v) for v in kt.values()) / 10)
        res[name]["gpu_ms_fwd_bwd"] = round(float(np.median(blocks_ms)), 4)
        res[name]["launch_calls"] = {k: len(v) // 10 for k, v in kt.items()}
        # The roofline of every row is taken over the step replayed as ONE HIP graph back to back (graph_step_ms: median of five blocks of
        # 50 replays after a ramp): GPU-bound whatever the batch, no event pair between kernels, comparable across boxes. For the
        # large-batch rows it agrees with the eager wall time (the host runs ahead of the GPU there); for the reference's training batches
        # the eager wall is the host's. `traffic` = HBM bytes per step from the PMC counters where a counter file of these kernels exists
        # (profiles/rNN/hbm_traffic_secondary.json).
        tg = graph_step_ms(step)
        res[name]["ms_fwd_bwd_graph"] = tg
        a = bytes_per_cs * cs / (tg * 1e-3) / 1e9
        res[name]["roofline"] = {"bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4),
                                 "algorithmic_bytes": int(bytes_per_cs * cs), "traffic": secondary_traffic(name), "over": "ms_fwd_bwd_graph"}
        if note:
            res[name]["note"] = note
        del x, w

    ctl1 = lambda lo, hi: (lambda B: (rnd(B) * (hi - lo) + lo).requires_grad_(True))
    bench_op("gain", 256, 2, 131072, lambda B: ([ctl1(-24, 24)(B)], lambda x, c: D.gain(x, SR, c[0])), 20)
    bench_op("distortion", 256, 2, 131072, lambda B: ([ctl1(0, 24)(B * 2)], lambda x, c: D.distortion(x, SR, c[0])), 20)
    rng = [(-60, 0), (1, 20), (5, 100), (5, 100), (1e-3, 12), (0, 12)]
    bench_op("compressor", 256, 2, 262144, lambda B: ([ctl1(lo, hi)(B) for lo, hi in rng], lambda x, c: D.compressor(x, SR, *c)), 20)

    bench_op("expander", 256, 2, 262144, lambda B: ([ctl1(lo, hi)(B) for lo, hi in rng], lambda x, c: D.expander(x, SR, *c)), 20,
             "BASELINE config 3 names it beside the compressor; the reference's expander is a stub (functional.py:402-403): mode 1 of the same kernels")

    def speechlike(B, C, N):   # white noise x a slow random envelope spanning -60 .. 0 dBFS (all three knee regions)
        knots = rnd(B, 1, N // 4096 + 2) * -60.0
        env_db = torch.nn.functional.interpolate(knots, size=N, mode="linear", align_corners=True)
        return (rnd(B, C, N) * 2 - 1) * torch.pow(10.0, env_db / 20.0)
    bench_op("compressor_speechlike", 256, 2, 262144, lambda B: ([ctl1(lo, hi)(B) for lo, hi in rng], lambda x, c: D.compressor(x, SR, *c)), 20,
             "same op on white noise x slow random envelope, -60 .. 0 dBFS", xmake=speechlike)
    bench_op("noise_shaped_reverberation", 128, 2, 262144,
             lambda B: ([ctl1(0, 1)(B) for _ in range(25)], lambda x, c: D.noise_shaped_reverberation(x, SR, *c, device_noise=True)),
             2 * 0.537e9 / (128 * 2 * 262144),
             "noise generated inside the filter-bank kernels (device_noise=True): algorithmic bytes = x, y, gy, gx only (2 x 0.537 GB; SURVEY 8(d) "
             "with the noise terms dropped). With the noise counted as an input (2 x 1.354 GB, the round-2 convention) multiply frac by 2.52")
    # the headline op as the reference's chain calls it (examples/style_transfer.py:150: first effect, its input needs no gradient) and
    # at the reference's training batch sizes (examples/style_transfer.py:403, auto_eq.py:231), where rows are cut into segments
    peq = lambda B: ([ctl1(lo, hi)(B) for lo, hi in PEQ_RANGES], lambda x, c: D.parametric_eq(x, SR, *c))
    bench_op("parametric_eq_controls_only", 256, 2, 131072, peq, 16, "no gradient for x (8 B fwd + 8 B bwd per channel-sample)", x_grad=False)
    bench_op("parametric_eq_b16", 16, 2, 131072, peq, 20, "reference training batch: segmented rows")
    bench_op("compressor_b8", 8, 2, 262144, lambda B: ([ctl1(lo, hi)(B) for lo, hi in rng], lambda x, c: D.compressor(x, SR, *c)), 20,
             "reference training batch")
    bench_op("noise_shaped_reverberation_b8", 8, 2, 131072,
             lambda B: ([ctl1(0, 1)(B) for _ in range(25)], lambda x, c: D.noise_shaped_reverberation(x, SR, *c, device_noise=True)),
             2 * 0.537e9 / (128 * 2 * 262144), "reference training batch: the filter bank's bands dealt out over workgroups")
    # The drop-in DEFAULT of noise_shaped_reverberation: the reference draws torch.randn(2 bs, 12, 66558) from the global CPU generator per
    # call (functional.py:548) - same torch.manual_seed, same impulse responses. Until round 5 this package drew it the same way (one host
    # thread + a host-to-device copy in front of the kernels: 682 ms per step at (128,2,262144)); since round 6 the same stream - values and
    # generator state - is computed on the device from the generator's state (csrc/mtrand.hip). Wall time per fwd+bwd step of the default
    # call, the GPU time of the stream's kernels alone, and the host draw + copy it replaced timed beside it (one step with the device
    # stream switched off).
    def default_noise_wall(name, B, N, steps):
        from dasp_pytorch_amd import _mt19937
        x = (rnd(B, 2, N) * 2 - 1).requires_grad_(True)
        ctl = [ctl1(0, 1)(B) for _ in range(25)]
        w = torch.randn(B, 2, N, device=dev, generator=g)

        def step():
            x.grad = None
            for c in ctl:
                c.grad = None
            D.noise_shaped_reverberation(x, SR, *ctl).backward(w)
        t = float(np.median([_time_steps(step, steps=steps, warmup=2 if i == 0 else 0) for i in range(3)]))
        size = (B * 2, 12, 65536 + 1023 - 1)
        gen = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            noise = _mt19937.randn_cpu_stream(*size, device=dev)
            e1.record()
            torch.cuda.synchronize()
            gen.append(e0.elapsed_time(e1))
        _mt19937.enabled = False
        try:
            t_host_step = _time_steps(step, steps=1, warmup=0)
        finally:
            _mt19937.enabled = True
        t0 = time.perf_counter()
        noise = torch.randn(*size)
        t_rng = time.perf_counter() - t0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        noise.to(dev)
        torch.cuda.synchronize()
        t_h2d = time.perf_counter() - t0
        res[name] = {"shape": [B, 2, N], "ms_fwd_bwd_wall": round(t * 1e3, 3), "noise_stream_gpu_ms": round(float(np.median(gen)), 3),
                     "noise_values": int(noise.numel()), "noise_bytes": int(noise.numel() * 4),
                     "host_draw": {"ms_fwd_bwd_wall": round(t_host_step * 1e3, 2), "host_randn_ms": round(t_rng * 1e3, 2), "h2d_copy_ms": round(t_h2d * 1e3, 2),
                                   "host_threads": torch.get_num_threads()},
                     "note": "default (reference-compatible) noise: torch's CPU random stream (MT19937, 24-bit floats, 16-wide Box-Muller layout) computed on "
                             "the device from the CPU generator's state, generator state written back (csrc/mtrand.hip; tests/test_gpu_mtrand.py: equal to "
                             "torch.randn to 1e-6, state bit-equal); host_draw = the same step with the noise drawn by torch.randn on the host and copied, "
                             "as rounds 1-5 did and the reference does"}
        del x, w, noise
    if not skip_default_noise:
        default_noise_wall("noise_shaped_reverberation_default_noise", 128, 262144, 3)
        default_noise_wall("noise_shaped_reverberation_default_noise_b8", 8, 131072, 10)
    # widening rows: stereo utilities and the multi-resolution STFT loss
    bench_op("stereo_widener", 256, 2, 131072, lambda B: ([ctl1(0, 1)(B)], lambda x, c: D.stereo_widener(x, SR, c[0].reshape(-1, 1))), 20)
    # the boundary's long tail: lfilter_via_fsm with more than three coefficients (signal.py:95-133; csrc/lfilter.hip: double arithmetic,
    # chunks of time side by side - not one of the tuned kernels, timed so that the record is complete)
    import scipy.signal as _ss
    _ba = _ss.butter(4, 0.3)
    lf_b = torch.tensor(np.tile(_ba[0], (16, 1)), dtype=torch.float32, device=dev).requires_grad_(True)
    lf_a = torch.tensor(np.tile(_ba[1], (16, 1)), dtype=torch.float32, device=dev).requires_grad_(True)
    lf_x = (rnd(16, 1, 262144) * 2 - 1).requires_grad_(True)
    lf_w = torch.randn(16, 1, 262144, device=dev, generator=g)

    def lf_step():
        lf_x.grad = None; lf_b.grad = None; lf_a.grad = None
        D.signal.lfilter_via_fsm(lf_x, lf_b, lf_a).backward(lf_w)
    res["lfilter_via_fsm_k5_b16"] = {"shape": [16, 1, 262144], "coefficients": 5, "ms_fwd_bwd": round(_time_steps(lf_step, steps=10, warmup=3) * 1e3, 3),
                                     "note": "4th-order Butterworth per row, gradients for x, b, a; recurrence in double arithmetic"}
    del lf_x, lf_w
    # the reference's whole effect chain as its training loop calls it (examples/style_transfer.py:150-154: EQ -> compressor -> reverb -> gain
    # through process_normalized, mono input, no gradient for it) at its batch size: everything of SURVEY 8(f) that is in - fused
    # process_normalized, no-gx EQ backward, segmented rows / items, gain folded into the make-up gain, no saved wet signal
    chain = D.chain.StyleTransferChain(SR, device_noise=True)
    xc = rnd(16, 1, 131072) * 2 - 1
    pcs = [(rnd(16, n) * 0.9 + 0.05).requires_grad_(True) for n in chain.num_params]
    wc = torch.randn(16, 2, 131072, device=dev, generator=g)

    def chain_step():
        for p in pcs:
            p.grad = None
        chain.process_normalized(xc, *pcs).backward(wc)
    t = _time_steps_median(chain_step)
    _lib.timers.start(every=1)
    for _ in range(10):
        chain_step()
    kt = _lib.timers.stop()
    # the same step with forward and backward replayed as HIP graphs inside an ordinary autograd step (torch.cuda.make_graphed_callables;
    # the noise seed is then a fixed base plus a device word the caller bumps): what an eager training loop can have without its host time
    off = torch.zeros(1, dtype=torch.int64, device=dev)
    chain_g = D.chain.StyleTransferChain(SR, device_noise=True, noise_seed=7, noise_seed_offset=off)
    graphed = torch.cuda.make_graphed_callables(lambda x_, a, b, c, d: chain_g.process_normalized(x_, a, b, c, d),
                                                (xc.clone(),) + tuple(p.detach().clone().requires_grad_(True) for p in pcs))

    def chain_step_graphed():
        for p in pcs:
            p.grad = None
        off.add_(1)
        graphed(xc, *pcs).backward(wc)
    tg = _time_steps(chain_step_graphed)
    # box-independent GPU time: the whole training-like step as one HIP graph, replayed back to back (graph_step_ms)
    def chain_step_static():
        for p in pcs:
            p.grad = None
        chain_g.process_normalized(xc, *pcs).backward(wc)
    t_graph = graph_step_ms(chain_step_static)
    # eager wall time by binding: torch.ops.dasp.* (C++ autograd, csrc/torch_ext) against the ctypes autograd.Functions, and without the
    # [0, 1] range check (one host read-back per step: the host cannot run ahead of the GPU across it)
    from dasp_pytorch_amd import _torch_ops
    eager = {"torch_ops" if _torch_ops.enabled() else "ctypes": round(t * 1e3, 3)}
    for proc in (chain.equalizer, chain.compressor, chain.reverb, chain.gain):
        proc.validate_range = "deferred"              # the check stays, read one call late from pinned memory: no host wait
    eager[("torch_ops" if _torch_ops.enabled() else "ctypes") + "_deferred_range_check"] = round(_time_steps_median(chain_step) * 1e3, 3)
    chain.flush_range_check()
    for proc in (chain.equalizer, chain.compressor, chain.reverb, chain.gain):
        proc.validate_range = False
    eager[("torch_ops" if _torch_ops.enabled() else "ctypes") + "_no_range_check"] = round(_time_steps_median(chain_step) * 1e3, 3)
    if _torch_ops.enabled():
        with D.config.override(torch_ops=False):
            eager["ctypes_no_range_check"] = round(_time_steps_median(chain_step) * 1e3, 3)
        cm = D.chain.ChainModule(SR, noise_seed=7, device=dev)

        def module_step():
            for p in pcs:
                p.grad = None
            cm(xc, *pcs).backward(wc)
        eager["chain_module_torch_ops"] = round(_time_steps_median(module_step) * 1e3, 3)
    res["style_transfer_chain_b16"] = {"shape": [16, 1, 131072], "ms_fwd_bwd": round(t * 1e3, 3),
                                       "ms_fwd_bwd_graphed_callable": round(tg * 1e3, 3),
                                       "ms_fwd_bwd_graph": t_graph, "eager_ms_by_binding": eager,
                                       "eager_over_gpu": round(min(eager.values()) / t_graph, 3),
                                       "gpu_ms_fwd_bwd": round(sum(sum(v) for v in kt.values()) / 10, 4),
                                       "library_calls_per_step": sum(len(v) for v in kt.values()) // 10,
                                       "note": "EQ -> compressor -> reverb -> gain on normalised parameters, gradients for all 50 of them"}
    xs = (rnd(16, 2, 131072) * 0.6 - 0.3).requires_grad_(True)
    ys = rnd(16, 2, 131072) * 0.6 - 0.3
    loss_fn = D.losses.MultiResolutionSTFTLoss()

    def loss_step():
        xs.grad = None
        loss_fn(xs, ys).backward()
    t = _time_steps(loss_step)
    _lib.timers.start(every=1)
    for _ in range(10):
        loss_step()
    gpu_ms = sum(sum(v) for v in _lib.timers.stop().values()) / 10
    res["mrstft_loss"] = {"shape": [16, 2, 131072], "ms_fwd_bwd": round(t * 1e3, 3), "gpu_ms_fwd_bwd": round(gpu_ms, 4),
                          "channel_samples_per_s": 16 * 2 * 131072 / t, "roofline": mrstft_roofline(gpu_ms),
                          "note": "3 resolutions (1024/120/600, 2048/240/1200, 512/50/240); compute-bound (7.4 transforms per input sample "
                                  "and direction), HBM traffic is the two signals and the gradient; parity unpinned (auraloss absent)"}
    # the reference's target synthesis (examples/style_transfer.py:293-299: the chain without gradients, every training step): EQ + compressor
    # as one fused pass (csrc/chainfwd.hip) against the two separate forward calls
    from dasp_pytorch_amd import ops as _ops
    from dasp_pytorch_amd.functional import _PEQ_TYPES
    elo = [float(r[0]) for r in PEQ_RANGES]
    espan = [float(r[1] - r[0]) for r in PEQ_RANGES]
    dlo = torch.tensor([r[0] for r in rng], device=dev)
    dhi = torch.tensor([r[1] for r in rng], device=dev)
    for B, C, N in ((256, 2, 131072), (16, 1, 262144)):
        xf = rnd(B, C, N) * 2 - 1
        pn = rnd(B, 18)
        comp = rnd(B, 6) * (dhi - dlo) + dlo
        ctl = torch.cat([comp[:, :3], comp[:, 4:]], 1).contiguous()
        eqm = D.ParametricEQ(SR)
        eqm.validate_range = False
        ccols = [comp[:, i].contiguous() for i in range(6)]
        out = {}
        with torch.no_grad():
            for tag, fn in (("separate", lambda: D.compressor(eqm.process_normalized(xf, pn), SR, *ccols)),
                            ("fused", lambda: _ops.chain_eq_compressor_forward(xf, pn, _PEQ_TYPES, elo, espan, float(SR), ctl))):
                tw = _time_steps(fn)
                _lib.timers.start(every=1)
                for _ in range(10):
                    fn()
                out[tag] = {"ms": round(tw * 1e3, 4), "gpu_ms": round(sum(sum(v) for v in _lib.timers.stop().values()) / 10, 4)}
        cs = B * C * N
        a = 8 * cs / (out["fused"]["gpu_ms"] * 1e-3) / 1e9
        res[f"chain_forward_eq_compressor_b{B}"] = {"shape": [B, C, N], "separate": out["separate"], "fused": out["fused"],
                                                    "gpu_ratio_fused_over_separate": round(out["fused"]["gpu_ms"] / out["separate"]["gpu_ms"], 3),
                                                    "roofline": {"bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                                 "frac": round(a / HBM_PEAK_GBS, 4), "algorithmic_bytes": 8 * cs, "traffic": None},
                                                    "note": "forward only (no grad): 8 B per channel-sample fused, 16 B as two calls"}
        del xf
    return res


def load_traffic(shape):
    """HBM bytes per launch from the newest profiles/r*/hbm_traffic.json whose shape and kernel-source hash match what is loaded."""
    want = kernel_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic.json")), reverse=True):
        try:
            tj = json.load(open(path))
            if tj.get("shape") == list(shape) and tj.get("kernel_source_hash") == want:
                t = {"fwd": tj["sos_fwd_kernel"]["hbm_bytes"], "bwd": tj["sos_bwd_kernel"]["hbm_bytes"]}
                t["both"] = t["fwd"] + t["bwd"]
                return t, os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    print(f"bench.py: no profiles/r*/hbm_traffic.json matches shape {list(shape)} and kernel source hash {want}: roofline.traffic is null "
          "(re-run scripts/hbm_traffic.sh on the GPU box)", file=sys.stderr)
    return {}, None


def profiled_kernel_ms():
    """Average durations (ms) of the two EQ kernels from the newest committed rocprofv3 --kernel-trace --stats summary of this command
    (profiles/rNN/bench_kernel_stats.csv), for the record next to the live HIP-event figures: the profiler sees the kernels inside the
    uninstrumented step, the event pass inside a four-call instrumented one, and the two have differed by 2 - 4 % (events read the
    power-limited backward kernel fast)."""
    import csv
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_kernel_stats.csv")), reverse=True):
        try:
            out = {}
            for r in csv.DictReader(open(path)):
                for key, names in (("bwd", ("sos_bwd_gram_kernel<6", "sos_bwd_kernel<6")), ("fwd", ("sos_fwd_kernel<6",))):
                    if any(n in r["Name"] for n in names) and key not in out:
                        out[key] = float(r["AverageNs"]) / 1e6
            if len(out) == 2:
                return {"bwd": out["bwd"], "fwd": out["fwd"], "file": os.path.relpath(path, ROOT)}
        except (OSError, KeyError, ValueError):
            continue
    return None


def _respawn_under_torchrun(n):
    """`python bench.py --gpus N` outside a torchrun environment: re-launch this command line as N ranks on this node, one per GPU
    (the launch line the driver uses for N > 1), so that --gpus N cannot silently be a one-rank run."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a torchrun environment: launching %s" % (n, " ".join(cmd)), file=sys.stderr)
    raise SystemExit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--blocks", type=int, default=5,
                    help="timed blocks of exactly --steps steps each (every block bracketed by barrier + device sync, max over ranks); "
                         "`value` / `ms_per_step` are the median block, `block_ms_per_step` has min / median / max per launch mode")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch items per GPU; strong: --batch items in total, partitioned over the GPUs (SURVEY 8e)")
    ap.add_argument("--launch", choices=("eager", "graph", "both"), default="both",
                    help="how the step is issued: from Python every step, as a replayed HIP graph of the same launches, or both (default; "
                         "value = the faster, config.launch names it, launch_ms_per_step has both)")
    ap.add_argument("--ramp-seconds", type=float, default=1.0,
                    help="untimed clock ramp before each launch mode's warmup steps, issued in that mode: the MI355X needs ~0.2 s of "
                         "sustained load to leave its idle clocks (measured: the same kernels run 1.28x slower in the first 10 ms)")
    ap.add_argument("--batch", type=int, default=256, help="batch items per GPU (weak) / in total (strong)")
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--samples", type=int, default=131072)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="time the reference on the WHOLE (256,2,131072) workload (minutes, ~14 GB of host memory) instead of its best-case (8,2,131072) sub-batch")
    ap.add_argument("--no-default-noise-rows", action="store_true", help="developer switch: skip the secondary rows of the reverb's default noise")
    ap.add_argument("--no-event-backlog", action="store_true", help="developer switch: no backlog of graph replays in front of the per-kernel event pass")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short timings of the other hot-path ops")
    ap.add_argument("--no-kernel-events", action="store_true", help="developer switch: skip the per-kernel HIP-event pass (roofline is then NaN)")
    ap.add_argument("--dry-run-cpu", action="store_true", help="test hook: rank wiring on CPU tensors over gloo, kernels replaced by a copy")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _respawn_under_torchrun(args.gpus)
    rank, local, world = dd.env_world()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to report a "
                         f"{args.gpus}-GPU number from {world} process(es)")
    dry = args.dry_run_cpu
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists in dasp_pytorch_amd)")
    if dry:
        dev = torch.device("cpu")
    else:
        if torch.cuda.device_count() <= local:
            raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but this node shows {torch.cuda.device_count()} device(s)")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist = None
    # one rank started by torch.distributed.run (the driver's launch line with --nproc-per-node 1) joins a one-rank process group too:
    # barrier and max-over-ranks then go through RCCL, which is all of the N > 1 path a 1-GPU box can exercise
    under_launcher = world == 1 and bool(os.environ.get("TORCHELASTIC_RUN_ID"))
    if world > 1 or under_launcher:
        import torch.distributed as dist
        dd.init("gloo" if dry else "nccl", None if dry else dev, force=under_launcher)
    sync = (lambda: None) if dry else torch.cuda.synchronize
    joined = dist.get_world_size() if dist is not None else 1      # ranks that actually joined the process group (RCCL / gloo)

    C, N = args.channels, args.samples
    if args.scaling == "strong":      # the one global batch, contiguous shards (distributed.shard_bounds)
        lo, hi = dd.shard_bounds(args.batch, world, rank)
        B, global_batch = hi - lo, args.batch
        if B == 0:
            raise SystemExit(f"--scaling strong: rank {rank} of {world} owns no item of a batch of {args.batch}")
        x, params, w = make_batch(args.batch, C, N, 1234, "cpu")
        x, params, w = (t[lo:hi].contiguous().to(dev) for t in (x, params, w))
    else:
        B, global_batch = args.batch, args.batch * world
        x, params, w = make_batch(B, C, N, 1234 + rank, dev)
    # items per rank, as every rank reports them (the shards of a strong-scaling run may differ by one item; they must add up)
    shard_items = [B]
    if dist is not None:
        got = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(joined)]
        dist.all_gather(got, torch.tensor([B], dtype=torch.int64, device=dev))
        shard_items = [int(t.item()) for t in got]
        if sum(shard_items) != global_batch:
            raise SystemExit(f"bench.py: the ranks own {shard_items} items, which is not the global batch of {global_batch}")
    x.requires_grad_(True)
    cols = [params[:, i].clone().requires_grad_(True) for i in range(18)]
    peq = (lambda x, sr, *c: x * 1.0 + 0.0 * sum(c)[:, None, None]) if dry else D.parametric_eq

    def step():
        x.grad = None
        for c in cols:
            c.grad = None
        y = peq(x, SR, *cols)
        y.backward(w)

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
            sync()

    def timed(fn, steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        fence()
        return dd.max_over_ranks(time.perf_counter() - t0, dev)

    def ramp(fn, seconds):
        """Untimed sustained load issued the way the timed steps will be (the queue is drained every 25 steps so that it stays short)."""
        t0 = time.perf_counter()
        while not dry and time.perf_counter() - t0 < seconds:
            for _ in range(25):
                fn()
            sync()

    def measure(fn):
        """Ramp, W warm-up steps, then --blocks blocks of exactly K steps; the product path only: no timers, no events, no extra calls."""
        ramp(fn, args.ramp_seconds)
        for _ in range(args.warmup):
            fn()
        return [timed(fn, args.steps) for _ in range(max(1, args.blocks))]

    assert not _lib.timers.enabled
    blocks = {}
    if args.launch in ("eager", "both") or dry:
        blocks["eager"] = measure(step)
    if args.launch in ("graph", "both") and not dry:
        # the same step captured once (torch.cuda.graph: forward + backward on the capture stream, gradients land in static buffers)
        # and replayed: no Python, ctypes or autograd time per step, launch gaps are the graph's
        x.grad = None
        for c in cols:
            c.grad = None
        sync()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y = peq(x, SR, *cols)
            y.backward(w)
        blocks["graph"] = measure(graph.replay)
    ktimes = {}
    if not args.no_kernel_events and not dry:
        # per-kernel durations: a separate, untimed pass right behind the timed blocks (warm clocks) with HIP events on the launch stream
        # around every C entry point. With the timers on, the ops issue design / cascade / adjoint / finalize as separate entry points
        # (the same kernels the one-call product path launches), so that each kernel gets its own pair of events.
        # Events around every 4th launch of each entry point only, over 240 back-to-back steps: the chip runs these kernels at its power
        # limit, and the idle microseconds an event pair inserts let the next kernel run at a higher clock - with events around every
        # launch the kernels measured 4 % faster than rocprofv3 sees them in the uninstrumented step (profiles/r03/README.md).
        # The instrumented step costs more host time than the GPU needs for it (four C calls, autograd and event records: 0.47 - 0.62 ms
        # issued per step on the boxes measured, against 0.40 ms of kernels): left alone the pass is host-bound, the GPU waits for
        # launches, and a power-limited kernel runs at a higher clock after every wait (seen: backward events 3 - 11 % below the timed
        # step, by box). So the GPU is given a backlog first - graph replays, which cost the host ~15 us each - sized from a probe of the
        # host's issue rate so that the queue never runs dry while the 240 instrumented steps are issued behind it.
        ramp(step, 0.25)
        _lib.timers.start(every=4)
        sync()
        t_p0 = time.perf_counter()
        for _ in range(16):
            step()
        host_per_step = (time.perf_counter() - t_p0) / 16
        sync()
        _lib.timers.start(every=4)                                     # (the probe's samples are dropped)
        gpu_per_step = min(float(np.median(v)) for v in blocks.values()) / args.steps
        n_rep = 0
        if "graph" in blocks and not args.no_event_backlog:
            n_rep = min(4000, 200 + int(3.0 * 240 * max(0.0, host_per_step - gpu_per_step) / gpu_per_step))      # 3x the computed deficit: the probe ran on an idle host
            for _ in range(n_rep):
                graph.replay()
        for _ in range(240):
            step()
        ktimes = _lib.timers.stop()
        print(f"[bench] event pass: {1e3 * host_per_step:.3f} ms of host time per instrumented step, {1e3 * gpu_per_step:.3f} ms of GPU time; "
              f"{n_rep} graph replays queued in front", file=sys.stderr)
    finite = bool(torch.isfinite(x.grad).all().item()) and all(bool(torch.isfinite(c.grad).all().item()) for c in cols)
    # In-situ durations of the two passes of the step (round 4). The per-entry-point events above sit inside a four-call instrumented step
    # issued through autograd, whose extra launches and records leave the power-limited kernels a cooler chip: they read the backward kernel
    # 2 - 8 % faster than rocprofv3 sees it in the product step (kernel_events_over_step says so). Here the step is the product's own two C
    # calls - dasp_peq_forward (design + forward kernel), dasp_sosfilt_backward_grads_ex (adjoint kernel + finalize) - issued straight from
    # this loop (no autograd: ~0.05 ms of host time per step against ~0.39 ms of kernels, so the queue never runs dry), with ONE event
    # between the two calls and one at either end of every 4th step: nothing sits between the kernels of a pass, and the two intervals add
    # up to the step.
    pass_ms = None
    if not dry and not args.no_kernel_events:
        import ctypes
        from dasp_pytorch_amd import ops as _ops
        from dasp_pytorch_amd._lib import call as _call, ptr as _ptr, stream as _stream
        from dasp_pytorch_amd.functional import _PEQ_TYPES
        x32 = x.detach()
        c32 = [c.detach().contiguous() for c in cols]
        wk = _ops._SosWork(B, 6, x32, True)
        if not wk.tseg:
            rows = (ctypes.c_void_p * 18)(*[c.data_ptr() for c in c32])
            tys = (ctypes.c_int * 6)(*_PEQ_TYPES)
            ybuf = torch.empty_like(x32)
            evs = []

            def cstep(rec=False):
                if rec:
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    e[0].record()
                _call("dasp_peq_forward", rows, B, 6, tys, float(SR), _ptr(wk.tab), _ptr(wk.dtab), _ptr(x32), _ptr(ybuf), _ptr(wk.carries), B, C, N, 0,
                      _ptr(None), _ptr(None), _stream())
                if rec:
                    e[1].record()
                wk.backward(x32, w, 2, 1, True, True)
                if rec:
                    e[2].record()
                    evs.append(e)
            ramp(cstep, 0.5)
            for i in range(240):
                cstep(rec=(i % 4 == 3))
            sync()
            pass_ms = {"forward_pass": float(np.mean([e[0].elapsed_time(e[1]) for e in evs])),
                       "backward_pass": float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))}
            del ybuf, wk

    if rank == 0:
        units = B * C * N                       # channel-samples per step on this GPU
        total_units = global_batch * C * N      # ... over the whole job (weak: N x units; strong: the one batch)
        med = {k: float(np.median(v)) for k, v in blocks.items()}
        mode = min(med, key=med.get)
        dt = med[mode]
        ms = dt / args.steps * 1e3
        value = total_units / (dt / args.steps)
        nan = float("nan")
        t_fwd_iso = float(np.mean(ktimes.get("dasp_sosfilt_forward", [nan]))) * 1e-3        # events around the entry points of the instrumented step
        t_bwd_iso = float(np.mean(ktimes.get("dasp_sosfilt_backward_ex", [nan]))) * 1e-3
        t_small = sum(float(np.mean(v)) for k, v in ktimes.items() if k not in ("dasp_sosfilt_forward", "dasp_sosfilt_backward_ex")) * 1e-3
        t_prep = float(np.mean(ktimes.get("dasp_peq_prepare_rows", [nan]))) * 1e-3
        t_fin = float(np.mean(ktimes.get("dasp_sos_grad_finalize_ex", [nan]))) * 1e-3
        if pass_ms is not None and t_prep == t_prep and t_fin == t_fin:
            # the dominant kernels inside the product step: the pass they are the bulk of, minus the small kernel of that pass (design / finalize:
            # latency-bound, so their isolated event durations hold); the launch gap inside the pass stays with the kernel (conservative)
            t_fwd = pass_ms["forward_pass"] * 1e-3 - t_prep
            t_bwd = pass_ms["backward_pass"] * 1e-3 - t_fin
        else:
            t_fwd, t_bwd = t_fwd_iso, t_bwd_iso
        traffic, traffic_file = ({}, None) if dry else load_traffic((B, C, N))

        def roof(bytes_per_sample, t, which=None):
            a = bytes_per_sample * units / t / 1e9
            return {"bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4),
                    "traffic": traffic.get(which)}

        per_step = lambda v: round(v / args.steps * 1e3, 5)
        out = {
            "metric": "audio-samples/sec fwd+bwd, 6-band parametric_eq @ (256,2,131072)",
            "value": value, "unit": "channel-samples/s", "n_gpus": joined, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "ramp_s": args.ramp_seconds,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"parametric_eq fwd+bwd (grad x + 18 controls) on ({B},{C},{N}) fp32 per GPU, sr 44100, "
                                   "controls ~ U(ParametricEQ ranges)", "global_batch": global_batch,
                       "parallelism": f"batch-shard x{world}, no collective", "launch": mode, "world_size": joined, "shard_items": shard_items,
                       "process_group": (dist.get_backend() if dist is not None else None),
                       "timing": f"median of {len(blocks[mode])} blocks of {args.steps} steps, product path (no timers / events in the timed region)"},
            "launch_ms_per_step": {k: per_step(v) for k, v in med.items()},
            "block_ms_per_step": {k: {"min": per_step(min(v)), "median": per_step(float(np.median(v))), "max": per_step(max(v))}
                                  for k, v in blocks.items()},
            "roofline": dict(roof(12, t_bwd, "bwd"), kernel="sos_bwd_gram_kernel<6>", ms=round(t_bwd * 1e3, 4),
                             algorithmic_bytes=12 * units),
            "roofline_fwd": dict(roof(8, t_fwd, "fwd"), kernel="sos_fwd_kernel<6>", ms=round(t_fwd * 1e3, 4), algorithmic_bytes=8 * units),
            "roofline_fwd_bwd": dict(roof(20, t_fwd + t_bwd, "both"), ms=round((t_fwd + t_bwd) * 1e3, 4), algorithmic_bytes=20 * units),
            "small_kernels_ms": round(t_small * 1e3, 4),
            # how the two durations above were measured: in situ (the step as two graph replays with one event between them; roofline.ms =
            # the backward pass minus the finalize kernel's isolated duration, launch gap included) - and, for comparison, the isolated
            # per-entry-point event durations of the four-call instrumented step, which read power-limited kernels fast
            "passes_ms": None if pass_ms is None else {k: round(v, 4) for k, v in pass_ms.items()},
            "isolated_events_ms": {"fwd": round(t_fwd_iso * 1e3, 4), "bwd": round(t_bwd_iso * 1e3, 4), "design": round(t_prep * 1e3, 4), "finalize": round(t_fin * 1e3, 4)},
            # the same kernels' average durations in the newest committed rocprofv3 summary of this command (another run, possibly another
            # box; profiles/rNN/bench_kernel_stats.csv) and the roofline fractions they give on the same algorithmic bytes
            "rocprofv3_committed": (lambda pk: None if pk is None or (B, C, N) != (256, 2, 131072) else {
                "file": pk["file"], "bwd_ms": round(pk["bwd"], 4), "fwd_ms": round(pk["fwd"], 4),
                "bwd_frac": round(12 * units / (pk["bwd"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "fwd_frac": round(8 * units / (pk["fwd"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})(profiled_kernel_ms()),
            # consistency of the event pass with the timed step: the four kernels' event durations over the step (gaps between the kernels
            # are the rest; a ratio well below ~0.97 means the instrumented pass ran the kernels at another clock than the timed step did)
            "kernel_events_over_step": round(((pass_ms["forward_pass"] + pass_ms["backward_pass"]) * 1e-3 if pass_ms is not None else (t_fwd + t_bwd + t_small)) / (dt / args.steps), 4),
            # the same rate in frames (B N per step over all ranks) and the step as achieved HBM rate on its algorithmic bytes
            "frames_per_s": value / C,
            "step_algorithmic_GBps_per_gpu": round(20 * units / (dt / args.steps) / 1e9, 1),
            "traffic_file": traffic_file, "kernel_source_hash": kernel_source_hash(),
            "finite": finite,
        }
        if dry:
            out["dry_run"] = "CPU wiring test: kernels replaced by a copy, numbers are meaningless"
        if world == 1 and not args.no_secondary and not dry:
            out["secondary"] = secondary(dev, skip_default_noise=args.no_default_noise_rows)
        if world == 1 and not args.no_cpu_baseline and not dry:
            out["cpu_baseline"] = cpu_baseline(full=args.cpu_baseline_full)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
