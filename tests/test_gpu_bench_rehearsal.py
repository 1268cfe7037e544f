"""GPU: the N > 1 code path of bench.py end to end on a one-GPU box -- two ranks under torch.distributed.run, both on cuda:0 over gloo
(NGP_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device).  Timing lines of a rehearsal are not measurements; what is checked is that the
launch contract of the task holds: rank 0 prints ONE JSON line with the whole-job aggregate, n_gpus = 2, and the per-rank work is what a
single rank does (weak scaling), for rendering and for the training step with its gradient exchange."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(extra):
    env = dict(os.environ, NGP_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu"] + extra
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                            # rank 0 only, one line
    return json.loads(lines[0])


def test_two_rank_render_line():
    j = _run(["--steps", "3", "--warmup", "1", "--res", "200"])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["unit"] == "ray-samples/s"
    assert j["config"]["frames_per_gpu"] == 3 and j["value"] > 0
    # whole-job aggregate: both ranks' samples over the slower rank's time
    per_frame = j["config"]["samples_per_ray"] * j["config"]["rays_per_frame"]
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 / (2 * per_frame) - 1.0) < 0.2


def test_two_rank_training_line():
    j = _run(["--mode", "train", "--steps", "3", "--warmup", "1", "--settle", "2"])
    assert j["n_gpus"] == 2 and j["unit"] == "rays/s" and j["config"]["rays_per_step_per_gpu"] == 4096
    assert j["value"] > 0 and j["config"]["final_loss"] == j["config"]["final_loss"]      # finite loss after the exchanged steps


def _run_bare(extra):
    """the bare form the driver records (`python3 bench.py --gpus N ...`, no launcher, WORLD_SIZE unset): bench.py starts its own ranks"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["NGP_BENCH_REHEARSAL"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu"] + extra, env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]      # stdout is the one JSON line and nothing else
    return json.loads(lines[0])


def test_bare_command_starts_its_own_ranks_render():
    j = _run_bare(["--steps", "3", "--warmup", "1", "--res", "200"])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["value"] > 0


def test_bare_command_starts_its_own_ranks_train():
    j = _run_bare(["--mode", "train", "--steps", "3", "--warmup", "1", "--settle", "2"])
    assert j["n_gpus"] == 2 and j["unit"] == "rays/s" and j["value"] > 0


def test_bare_command_propagates_a_failing_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["NGP_BENCH_REHEARSAL"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu", "--scaling", "strong", "--path", "drop_in", "--steps", "1",
                          "--warmup", "0", "--res", "64"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]      # (strong scaling asserts path == fused)
