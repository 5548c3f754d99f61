"""Dry run of scripts/multi_gpu_bringup.py (the one command for the first multi-GPU node) on CPU over gloo at N = 2: the launches through
torch.distributed.run, the parsing of bench.py's line and of the bucket worker's, the stage bookkeeping and the exit code - and the RCCL
log parser on a captured line."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _module():
    spec = importlib.util.spec_from_file_location("multi_gpu_bringup", os.path.join(ROOT, "scripts", "multi_gpu_bringup.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_rccl_log_parser():
    m = _module()
    log = ("box:1:1 [0] NCCL INFO comm 0x55 rank 0 nranks 2 cudaDev 0 nvmlDev 0 busId c000 commId 0x1 - Init COMPLETE\n"
           "box:2:2 [1] NCCL INFO comm 0x66 rank 1 nranks 2 cudaDev 1 nvmlDev 1 busId e000 commId 0x1 - Init COMPLETE\n"
           "box:2:2 [1] NCCL INFO comm 0x66 rank 1 nranks 2 cudaDev 1 - Init START\n")
    assert m.rccl_ranks(log) == {0: 2, 1: 2}
    assert m.rccl_ranks("nothing here") == {}
    assert m.json_line('noise\n{"a": 1}\nmore\n{"n_gpus": 2}\n') == {"n_gpus": 2}


@pytest.mark.timeout(600)
def test_bringup_dry_run_world2(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "multi_gpu_bringup.py"), "--gpus", "2", "--dry-run-cpu", "--out", str(tmp_path),
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=560, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    s = json.load(open(tmp_path / "summary.json"))
    assert s["passed"] and s["dry_run"]
    names = [st["stage"] for st in s["stages"]]
    assert names == ["suite with two devices", "bench weak N=2", "bench strong N=2", "config 5 N=2"]
    c5 = s["stages"][3]
    assert c5["result"]["n_gpus"] == 2 and c5["result"]["buckets_launched_under_backward"] > 0 and c5["result"]["replicas_agree"]
    assert all(st["passed"] for st in s["stages"])
