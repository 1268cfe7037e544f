"""Host-side pieces of bench.py that need no GPU: the PSNR formula of the north_star criterion (nerf/utils.py:203-210, pinned by the PSNRMeter golden), the
garbage-collector guard around timed regions, the source-hash guard on profile counters."""
import gc
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
import bench  # noqa: E402


def test_psnr_ref_is_the_reference_formula():
    from ngp.metrics import PSNRMeter
    g = np.load(os.path.join(ROOT, "tests", "golden", "callers_tier1.npz"))
    p, t = g["psnr_pred0"], g["psnr_truth0"]
    assert bench.psnr_ref(p, t) == float(g["psnr_after0"])              # one update of the executed reference's PSNRMeter
    m = PSNRMeter(); m.update(torch.from_numpy(p), torch.from_numpy(t))
    assert bench.psnr_ref(p, t) == float(m.measure())


def test_timed_regions_run_without_gc_and_restore_it():
    assert gc.isenabled()
    with bench.no_gc_pauses():
        assert not gc.isenabled()
    assert gc.isenabled()
    gc.disable()
    try:
        with bench.no_gc_pauses():
            assert not gc.isenabled()
        assert not gc.isenabled()                                        # a caller that had it off keeps it off
    finally:
        gc.enable()


def test_committed_profiles_match_the_kernel_sources_of_this_tree():
    """roofline.traffic comes from profiles/rNN_pmc.csv only when that profile was collected for these kernel sources: the round's final profiles must match"""
    c, info = bench.committed_counters("r[0-9][0-9]_pmc.csv", bench.FRAME_SOURCES)
    assert c is not None and "FETCH_SIZE" in c, info
    c, info = bench.committed_counters("r[0-9][0-9]_train_pmc.csv", bench.TRAIN_SOURCES)
    assert c is not None and "WRITE_SIZE" in c, info


def _fake_launch(monkeypatch, returncode, stdout):
    import subprocess
    import types
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=returncode, stdout=stdout_text)
    stdout_text = stdout
    monkeypatch.setattr(subprocess, "run", fake_run)
    return seen


def test_bare_multi_gpu_command_launches_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus 4` with WORLD_SIZE unset (the form the driver records): one child launcher with N ranks on the loopback, the same
    arguments, rank 0's line relayed on stdout, everything else on stderr, exit 0 -- and no GPU call in this process"""
    import pytest
    seen = _fake_launch(monkeypatch, 0, 'noise from a rank\n{"metric": "m", "value": 1, "n_gpus": 4}\n')
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "m", "value": 1, "n_gpus": 4}' and "noise from a rank" in out.err
    assert not torch.cuda.is_initialized()


def test_bare_multi_gpu_command_fails_when_a_rank_does(monkeypatch, capsys):
    import pytest
    _fake_launch(monkeypatch, 3, "")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3 and capsys.readouterr().out == ""
