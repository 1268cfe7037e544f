"""GPU: RCCL itself, as far as one GPU allows (VERDICT r2 "Next" 3b).  A fresh child process initialises a one-rank `nccl` group on cuda:0
before any other GPU call and takes every collective of the N > 1 paths (barrier, float64 SUM / MAX, int64 + float32 all_gather, the 50.6 MB
in-place gradient all-reduce, the small bucket, a half payload, an async all-reduce).  This is SINGLE-RANK: it proves that RCCL loads,
that the calls are well-formed for device tensors of these dtypes and sizes, and that bench.py's --force-dist route works -- not scaling."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=600, env=None):
    e = dict(os.environ, **(env or {}))
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stdout[-2000:] + p.stderr[-2000:]
    return json.loads(lines[-1])


def test_rccl_single_rank_collectives(dev):
    out = _run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_single_rank.py")], env={"MASTER_PORT": "29547"})
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["reduce"] == [12345.0, 0.25]
    assert out["gather_equal"] and out["table_equal"] and out["bucket_equal"] and out["half_equal"] and out["int64_equal"] and out["async_equal"]
    assert out["table_bytes"] == 50630784


def test_bench_force_dist_takes_the_nccl_path(dev):
    """`bench.py --gpus 1 --force-dist`: the driver's own N = 1 command line with the process group on -- render line, then the training step
    with its gradient exchange"""
    line = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-fit"],
                env={"MASTER_PORT": "29548"})
    assert line["config"]["collectives"].startswith("nccl (RCCL), forced") and line["n_gpus"] == 1 and line["value"] > 1e8
    line = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--mode", "train", "--steps", "4", "--warmup", "2", "--settle", "4"],
                env={"MASTER_PORT": "29549"})
    assert line["unit"] == "rays/s" and line["value"] > 1e4 and line["config"]["final_loss"] == line["config"]["final_loss"]
