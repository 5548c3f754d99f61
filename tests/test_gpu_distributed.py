"""The multi-GPU path on the one GPU a test box has: RCCL (torch.distributed backend "nccl") initialised as a ONE-RANK group, so that the
collectives of BASELINE config 5 - the bucketed gradient all-reduce (hook-driven, overlapped with backward, and the post-backward form),
the barrier and the max-over-ranks of the timing - run through the backend on device buffers instead of being short-cut; bench.py under
`torch.distributed.run --nproc-per-node 1`; and the config-5 example with the reference-sized networks (10,327,346 parameters, 41.3 MB of
gradients in 16 MiB buckets). Each case runs in its own process (a process group is process-global state). Logs: gpurun_out/rccl_*.log."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    return env


def _log(name, text):
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", name), "w") as f:
            f.write(text)
    except OSError:
        pass


WORKER = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from dasp_pytorch_amd import distributed as dd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rank, world = dd.init("nccl", dev, force=True)
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    print("backend", dist.get_backend(), "world", dist.get_world_size(), "torch", torch.__version__, "device", torch.cuda.get_device_name(0))
    assert dd.max_over_ranks(1.25, dev) == 1.25                                   # an all-reduce(MAX) on a device scalar through RCCL
    dist.barrier(); torch.cuda.synchronize()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(256, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 50)).to(dev)
    x = torch.randn(32, 256, device=dev)
    net(x).square().mean().backward()
    want = [p.grad.clone() for p in net.parameters()]
    # post-backward form, forced through the backend on a one-rank group: sum over one rank / 1 = the same gradients
    nb = dd.allreduce_gradients(net.parameters(), bucket_bytes=1 << 20, force=True)
    assert nb >= 4 and all(torch.equal(p.grad, w) for p, w in zip(net.parameters(), want))
    # hook-driven form: flat buckets, launched from inside backward()
    gb = dd.GradientBuckets(net.parameters(), bucket_bytes=1 << 20, force=True)
    assert gb.active and len(gb.buckets) >= 4
    for it in range(3):
        gb.zero_grad()
        net(x).square().mean().backward()
        in_flight = gb._next
        n = gb.finish()
        torch.cuda.synchronize()
        assert n == len(gb.buckets) and in_flight >= 1, (n, in_flight)
        assert all(torch.allclose(p.grad, w, rtol=1e-6, atol=1e-9) for p, w in zip(net.parameters(), want))
    print("buckets", len(gb.buckets), "bytes", gb.bytes, "launched under backward", gb.launched_in_backward)
    gb.remove()
    dist.destroy_process_group()
    print("rccl one-rank ok")
""")


@pytest.mark.timeout(300)
def test_rccl_one_rank_group_runs_the_gradient_exchange():
    r = subprocess.run([sys.executable, "-c", WORKER % ROOT], capture_output=True, text=True, timeout=280,
                       env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
    _log("rccl_one_rank.log", r.stdout + "\n--- stderr ---\n" + r.stderr[-4000:])
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.timeout(600)
def test_bench_under_torchrun_one_process():
    """bench.py as the driver launches it for N > 1, with N = 1: torch.distributed.run, one rank, RCCL initialised (barrier + max over ranks
    through the backend), one JSON line with n_gpus = ranks that joined = 1."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--blocks", "2", "--no-secondary",
           "--no-cpu-baseline", "--ramp-seconds", "0.2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=560, env=_env())
    _log("rccl_bench_torchrun1.log", r.stdout + "\n--- stderr ---\n" + r.stderr[-4000:])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["world_size"] == 1 and out["config"].get("process_group") == "nccl" and out["finite"]
    assert out["value"] > 1e10


@pytest.mark.timeout(600)
@pytest.mark.parametrize("graph", [False, True])
def test_config5_reference_sized_networks(graph):
    """examples/style_transfer_synth.py --model reference: the reference's encoder / projector stack (10,327,346 parameters) in front of the
    chain, the step of examples/style_transfer.py:271-328 on synthetic clips, gradients exchanged through RCCL on a one-rank group in
    16 MiB buckets launched from inside backward() (eager) or after the graph replay (--graph)."""
    code = textwrap.dedent("""
        import importlib.util, json, sys
        spec = importlib.util.spec_from_file_location("sts", %r)
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        out, model = m.run(steps=3, batch=2, n=131072, ir_samples=16384, quiet=True, graph=%r, model_kind="reference", force_collectives=True)
        import torch
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        print("RESULT " + json.dumps(out))
    """) % (os.path.join(ROOT, "examples", "style_transfer_synth.py"), graph)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=560,
                       env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
    _log(f"rccl_config5_reference_graph{int(graph)}.log", r.stdout + "\n--- stderr ---\n" + r.stderr[-4000:])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert out["parameters"] == 10327346 and out["gradient_bytes_per_step"] == 4 * 10327346 and out["gradient_buckets"] == 3
    assert out["collectives_active"] and out["finite"]
    if not graph:
        assert out["buckets_launched_under_backward"] >= 2          # the projectors' and most of the encoder's buckets go out before backward returns
