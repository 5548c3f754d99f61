"""world_size-2 gloo tests of the multi-GPU plumbing (runs on CPU, no GPU needed): batch sharding covers the
batch exactly once, timing is the max over ranks, bucketed gradient all-reduce equals the mean of the ranks'
gradients, and bench.py's argument/launch contract parses."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest
import torch

from dasp_pytorch_amd import distributed as dd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    for n in (1, 2, 7, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [dd.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert dd.max_over_ranks(1.5) == 1.5 and dd.allreduce_gradients([torch.nn.Parameter(torch.ones(2))]) == 0
    # no process group: GradientBuckets still owns the gradient storage (flat buckets, .grad views) but issues nothing
    net = torch.nn.Linear(4, 3)
    gb = dd.GradientBuckets(net.parameters(), bucket_bytes=16)
    assert not gb.active and len(gb.buckets) == 2 and gb.bytes == 4 * 15
    gb.zero_grad(); net(torch.ones(2, 4)).sum().backward()
    assert gb.finish() == 0 and torch.equal(net.bias.grad, torch.full((3,), 2.0)) and net.bias.grad.data_ptr() == gb.buckets[0][0].data_ptr()
    gb.remove()


WORKER = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from dasp_pytorch_amd import distributed as dd
    rank, world = dd.init(backend="gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    # batch shards: disjoint, ordered, complete
    x = torch.arange(7 * 3, dtype=torch.float32).view(7, 3)
    mine = dd.shard_batch(x, world, rank)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mine.shape[0]]))
    assert sum(int(s) for s in sizes) == 7
    tot = mine.sum().clone(); dist.all_reduce(tot)
    assert float(tot) == float(x.sum())
    # timing: max over ranks
    assert dd.max_over_ranks(1.0 + rank) == 2.0
    # bucketed gradient all-reduce: mean over ranks, several buckets, a None grad on one rank only
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
    inp = torch.full((5, 16), float(rank + 1))
    net(inp).square().mean().backward()
    ref = [p.grad.clone() for p in net.parameters()]
    extra = torch.nn.Parameter(torch.ones(3))
    if rank == 0:
        extra.grad = torch.full((3,), 4.0)
    params = list(net.parameters()) + [extra]
    nb = dd.allreduce_gradients(params, bucket_bytes=1024)
    assert nb >= 2
    for p, g in zip(net.parameters(), ref):
        other = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(other, g)
        assert torch.allclose(p.grad, sum(other) / world, atol=1e-6)
    assert torch.allclose(extra.grad, torch.full((3,), 2.0))
    # the same exchange driven by post-accumulate-grad hooks (GradientBuckets: flat buckets autograd accumulates into, each all-reduced as
    # its last gradient arrives): equals the post-backward version above, bucket by bucket, over two steps (counters re-arm), with a
    # parameter that gets no gradient on either rank (its bucket is launched by finish()) and with .grad set to None by the caller
    torch.manual_seed(0)
    net2 = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
    unused = torch.nn.Parameter(torch.ones(5))
    gb = dd.GradientBuckets([unused] + list(net2.parameters()), bucket_bytes=256)       # (buckets fill in reverse order and launch in order: a parameter without a gradient in the FIRST bucket would hold every launch back until finish())
    assert gb.active and len(gb.buckets) >= 3 and gb.bytes == 4 * (sum(p.numel() for p in net2.parameters()) + 5)
    for it in range(2):
        if it == 1:
            net2[0].weight.grad = None                      # a caller's set_to_none: zero_grad() re-attaches the view
        gb.zero_grad()
        net2(inp).square().mean().backward()
        assert all(p.grad.data_ptr() == v.data_ptr() for _, members in gb.buckets for p, v in members)
        launched = gb._next                                 # buckets already in flight when backward() returned
        n = gb.finish()
        assert n == len(gb.buckets) and gb.launched_in_backward == launched and launched >= 1
        for p, q in zip(net2.parameters(), net.parameters()):
            assert torch.allclose(p.grad, q.grad, atol=1e-6), it
        assert float(unused.grad.abs().max()) == 0.0
    # one step = zero_grad -> backward -> finish (round 4, advisor): passes whose gradients are thrown away or only accumulated run under
    # no_sync() - nothing is launched from their hooks and the step's count stays where it was - and the step after them still exchanges
    # every bucket; a second backward() in one step raises from its first hook instead of all-reducing half-accumulated buckets
    with gb.no_sync():
        for _ in range(3):
            gb.zero_grad()
            net2(inp).square().mean().backward()
    assert gb._next == 0 and not gb._handles
    gb.zero_grad()
    net2(inp).square().mean().backward()
    assert gb.finish() == len(gb.buckets)
    for p, q in zip(net2.parameters(), net.parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-6)
    gb.zero_grad()
    with gb.no_sync():                                      # gradient accumulation: two local passes, one exchange of the sum
        net2(inp).square().mean().backward()
    net2(inp).square().mean().backward()
    assert gb.finish() == len(gb.buckets)
    for p, q in zip(net2.parameters(), net.parameters()):
        assert torch.allclose(p.grad, 2 * q.grad, atol=1e-5)
    gb.zero_grad()
    net2(inp).square().mean().backward()
    try:
        net2(inp).square().mean().backward()
        raise AssertionError("a second backward() before finish() must raise")
    except RuntimeError as e:
        assert "backward() ran twice" in str(e)
    gb.zero_grad()                                          # waits for what the aborted step left in flight, then a clean step again
    net2(inp).square().mean().backward()
    assert gb.finish() == len(gb.buckets)
    for p, q in zip(net2.parameters(), net.parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-6)
    gb.remove()
    dist.barrier()
    dist.destroy_process_group()
    print("worker", rank, "ok")
""")

# world 4: the ranks complete their buckets in different orders (on odd ranks a parameter of the FIRST bucket gets no gradient, so nothing can
# be launched from a hook there, while the even ranks launch every bucket under backward; rank 3 also sleeps inside backward): the
# collectives still pair up bucket by bucket, because every rank issues them in bucket order, and the result is the mean over the four ranks
WORKER4 = textwrap.dedent("""
    import os, sys, time, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from dasp_pytorch_amd import distributed as dd
    rank, world = dd.init(backend="gloo")
    assert world == 4
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2))
    tail = torch.nn.Parameter(torch.ones(4))                # LAST parameter = first bucket (buckets fill in reverse order)
    gb = dd.GradientBuckets(list(net.parameters()) + [tail], bucket_bytes=128)
    assert gb.active and len(gb.buckets) >= 4 and id(tail) in gb._bucket_of and gb._bucket_of[id(tail)] == 0
    if rank == 3:
        net[2].weight.register_hook(lambda g: (time.sleep(0.3), g)[1])
    for step in range(3):
        inp = torch.full((3, 8), float(rank + 1 + step))
        gb.zero_grad()
        out = net(inp).square().mean()
        if rank %% 2 == 0:
            out = out + (tail * (rank + 1)).sum()           # even ranks: every bucket completes under backward
        out.backward()
        launched = gb._next
        assert (launched == len(gb.buckets)) if rank %% 2 == 0 else (launched == 0), (rank, launched)
        mine = [p.grad.clone() for p in gb.params]          # (buckets already in flight hold partial sums: compare against a plain run)
        assert gb.finish() == len(gb.buckets)
        ref_net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2))
        ref_net.load_state_dict(net.state_dict())
        acc = [torch.zeros_like(p) for p in ref_net.parameters()]
        for r in range(world):
            ref_net.zero_grad()
            ref_net(torch.full((3, 8), float(r + 1 + step))).square().mean().backward()
            acc = [a + p.grad for a, p in zip(acc, ref_net.parameters())]
        for p, a in zip(net.parameters(), acc):
            assert torch.allclose(p.grad, a / world, atol=1e-6), (rank, step)
        assert torch.allclose(tail.grad, torch.full((4,), (1 + 3) / world)), tail.grad
    gb.remove()
    dist.barrier()
    dist.destroy_process_group()
    print("worker", rank, "ok")
""")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.timeout(180)
def test_world2_gloo():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=150)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"worker {rank} ok" in o, o


@pytest.mark.timeout(240)
def test_world4_gloo_uneven_bucket_completion():
    port = _free_port()
    procs = []
    for rank in range(4):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER4 % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=200)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"worker {rank} ok" in o, o


def test_bench_refuses_without_gpu_and_parses_contract():
    """bench.py must not silently fall back to a CPU path (there is none)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"], capture_output=True, text=True)
    assert r.returncode != 0 and "needs an MI355X" in (r.stderr + r.stdout)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_rank_wiring_under_torchrun(scaling):
    """bench.py as the driver launches it for N > 1 (python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 ...), on CPU over gloo with the kernels replaced by a copy (--dry-run-cpu): rendezvous, sharding (weak: every rank its
    own batch; strong: the one batch partitioned), barriers, max-over-ranks timing, and exactly one JSON line from rank 0 with the
    contract's keys."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--scaling", scaling, "--dry-run-cpu", "--batch", "6", "--samples", "2048"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == scaling and out["dry_run"]
    per_gpu = 6 if scaling == "weak" else 3
    assert out["config"]["global_batch"] == (12 if scaling == "weak" else 6)
    assert f"({per_gpu},2,2048)" in out["config"]["workload"]
    assert abs(out["value"] - out["config"]["global_batch"] * 2 * 2048 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_world8_as_the_driver_launches_it(scaling):
    """The driver's 8-GPU launch line (python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P
    bench.py --gpus 8 --steps K --warmup W) on CPU over gloo with the kernels replaced by a copy: eight ranks rendezvous, the batch is
    sharded (weak: 8 x the per-GPU batch; strong: one batch in 8 contiguous shards, here of unequal size), the timed region is the max
    over ranks, and rank 0 alone prints the line. No 8-GPU node has been available to any round: this is what can be verified without one."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--scaling", scaling, "--dry-run-cpu", "--batch", "20", "--samples", "1024", "--blocks", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=560, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["world_size"] == 8 and out["scaling"] == scaling and out["dry_run"]
    assert out["config"]["global_batch"] == (160 if scaling == "weak" else 20)
    assert abs(out["value"] - out["config"]["global_batch"] * 2 * 1024 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    if scaling == "strong":                                  # 20 items over 8 ranks: shards of 3 and 2 items, every item owned once
        assert out["config"]["shard_items"] == [3, 3, 3, 3, 2, 2, 2, 2]


@pytest.mark.timeout(300)
def test_bench_gpus_flag_means_ranks():
    """`python bench.py --gpus 2` without a torchrun environment starts two ranks itself (n_gpus = ranks that joined the process group);
    a launcher whose world size disagrees with --gpus is refused instead of reported as an N-GPU number."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "2", "--dry-run-cpu",
           "--batch", "4", "--samples", "1024"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world_size"] == 2 and out["config"]["global_batch"] == 8
    assert set(out["block_ms_per_step"]["eager"]) == {"min", "median", "max"}
    # world size 2 from the launcher, --gpus 1 on the command line: refuse
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--dry-run-cpu"],
                         capture_output=True, text=True, timeout=120,
                         env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
    assert bad.returncode != 0 and "refusing" in (bad.stderr + bad.stdout)
