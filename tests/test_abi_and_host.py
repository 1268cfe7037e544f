"""CPU: the C-ABI library loads without a GPU and exports every symbol include/ngp_hip.h declares; argument validation
returns error codes before anything touches a device; host-side logic (level tables, parameter layouts, workload
generator, ray generation, sharding helpers, PSNR) is checked against the formulas of the reference it mirrors."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ngp_hip
    header = open(os.path.join(ROOT, "include", "ngp_hip.h")).read()
    declared = set(re.findall(r"\b(ngp_[a-zA-Z0-9_]+)\s*\(", header)) - {"ngp_field_t"}
    assert len(declared) >= 28
    assert declared == set(ngp_hip.EXPORTS), declared ^ set(ngp_hip.EXPORTS)
    lib = ngp_hip.lib()                                           # dlopen works on a machine with no GPU
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ngp_abi_version() == 1
    assert ctypes.sizeof(ngp_hip.ngp_field_t) == 4 * 8 + 5 * 4 + 4   # 4 pointers, 5 scalars, tail padding


def test_argument_validation_needs_no_gpu():
    import ngp_hip
    L = ngp_hip.lib()
    one = ctypes.c_void_p(16)                                     # non-null dummy: validation rejects before any dereference
    assert L.ngp_grid_encode_forward(one, one, one, one, 4, 3, 3, 2, 1.0, 16, 0, None, 0, 0, 0, None) == -1
    assert b"C must be 1, 2, 4, or 8" in L.ngp_last_error()
    assert L.ngp_grid_encode_forward(one, one, one, one, 4, 7, 2, 2, 1.0, 16, 0, None, 0, 0, 0, None) == -1
    assert L.ngp_grid_encode_forward(one, one, one, one, 4, 3, 2, 40, 1.0, 16, 0, None, 0, 0, 0, None) == -1
    assert L.ngp_grid_encode_forward(one, one, one, one, 4, 3, 2, 2, 1.0, 16, 1, None, 0, 0, 0, None) == -1   # dy_dx missing
    assert L.ngp_sh_encode_forward(one, one, 4, 3, 9, 0, None, None) == -1
    assert b"degree in [1, 8]" in L.ngp_last_error()
    assert L.ngp_sh_encode_forward(one, one, 4, 2, 4, 0, None, None) == -1
    assert L.ngp_ffmlp_inference(one, one, 16, 32, 16, 128, 2, 0, 6, None, one, None) == -1
    assert L.ngp_ffmlp_inference(one, one, 16, 32, 3, 64, 2, 0, 6, None, one, None) == -1      # unpadded output width
    assert L.ngp_ffmlp_inference(one, one, 17, 32, 16, 64, 2, 0, 6, None, one, None) == -1      # batch not a multiple of 16
    assert L.ngp_ffmlp_inference(one, one, 16, 32, 16, 64, 2, 2, 6, None, one, None) == -1      # sine activation: unreachable in the reference
    assert L.ngp_near_far_from_aabb(None, None, None, 5, 0.2, None, None, None) == -1
    assert L.ngp_near_far_from_aabb(None, None, None, 0, 0.2, None, None, None) == 0            # empty input is fine
    assert L.ngp_march_rays(0, 4, None, None, None, None, 2.0, 0.0, 1024, 2, 128, None, None, None, None, None, None, 0, None) == 0
    assert L.ngp_march_rays_train_workspace(4096) >= 4 * (16 + 2)
    assert L.ngp_allocate_splitk(4) == 0 and L.ngp_free_splitk() == 0
    f = ngp_hip.ngp_field_t(16, 16, 16, 16, 8, 16, 0.5, 2.0, 1.0)                               # 8 levels: not the fused layout
    assert L.ngp_field_forward(ctypes.byref(f), one, one, 4, one, one, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import ngp_hip
    monkeypatch.setattr(ngp_hip, "_lib", None)
    monkeypatch.setattr(ngp_hip, "LIB_PATH", str(tmp_path / "libngp_hip.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        ngp_hip.lib()


def test_ops_refuse_cpu_tensors():
    """there is no CPU path: an op handed CPU tensors (on a box without a GPU) raises instead of computing"""
    from gridencoder import GridEncoder
    from shencoder import SHEncoder
    enc = GridEncoder(num_levels=2, log2_hashmap_size=8, desired_resolution=32)
    with pytest.raises(RuntimeError):
        enc(torch.zeros(4, 3))
    with pytest.raises(RuntimeError):
        SHEncoder()(torch.zeros(4, 3))


def test_grid_encoder_level_table_and_ffmlp_layout():
    from ffmlp import FFMLP
    from gridencoder import GridEncoder
    from ngp import workload as W
    for bound, rows in ((1, 6119864), (2, 6328848)):
        enc = GridEncoder(desired_resolution=2048 * bound)
        assert enc.embeddings.shape == (rows, 2) and enc.offsets.dtype == torch.int32 and enc.output_dim == 32
        assert float(enc.embeddings.abs().max()) <= 1e-4                # U(-1e-4, 1e-4) init (grid.py:133-135)
        off, pls = W.grid_offsets(bound)
        assert np.array_equal(off, enc.offsets.numpy()) and abs(pls - enc.per_level_scale) < 1e-12
    assert set(GridEncoder(num_levels=2, log2_hashmap_size=8, desired_resolution=32).state_dict()) == {"embeddings", "offsets"}
    net = FFMLP(32, 3, 64, 3)
    assert net.padded_output_dim == 16 and net.weights.shape == (64 * (32 + 128 + 16),)
    assert float(net.weights.abs().max()) <= math.sqrt(3 / 64) and list(net.state_dict()) == ["weights"]
    a, b = FFMLP(32, 16, 64, 2).weights.detach().clone(), FFMLP(32, 16, 64, 2).weights.detach().clone()
    assert torch.equal(a, b)                                            # torch.manual_seed(42) inside reset_parameters (ffmlp.py:141)
    with pytest.raises(AssertionError):
        FFMLP(31, 16, 64, 2)
    with pytest.raises(AssertionError):
        FFMLP(32, 17, 64, 2)
    # the reference's own limits (ffmlp.py:110-113); every shape inside them is accepted (width 64 / ReLU on the register-resident kernels,
    # the rest layer by layer: csrc/ffmlp_generic.hip)
    for ok in (dict(hidden_dim=128), dict(hidden_dim=16), dict(num_layers=5), dict(activation="sigmoid"), dict(input_dim=80)):
        kw = dict(input_dim=32, output_dim=16, hidden_dim=64, num_layers=2)
        kw.update(ok)
        net = FFMLP(**kw)
        assert net.weights.numel() == kw["hidden_dim"] * (kw["input_dim"] + kw["hidden_dim"] * (kw["num_layers"] - 1) + 16)
    for bad in (dict(hidden_dim=96), dict(hidden_dim=512), dict(num_layers=1), dict(input_dim=40)):
        kw = dict(input_dim=32, output_dim=16, hidden_dim=64, num_layers=2)
        kw.update(bad)
        with pytest.raises(AssertionError):
            FFMLP(**kw)


def test_workload_scene_model_and_bitfield(oracle):
    from ngp import workload as W
    grid = W.density_grid()
    assert grid.shape == (2, 128 ** 3) and set(np.unique(grid)) == {0.0, W.SIGMA_IN}
    bf, thresh = W.bitfield_from_grid(grid)
    assert 0 < thresh < 10 and np.array_equal(bf, oracle.packbits(grid, thresh))
    # the grid is conservative: every point inside a box lies in an occupied cell of every cascade that contains it
    rng = np.random.default_rng(0)
    boxes = W.scene_boxes()
    for lo, hi in boxes[::3]:
        p = rng.uniform(lo, hi, size=(200, 3))
        for cas in range(2):
            b = min(2.0 ** cas, W.BOUND)
            cell = np.clip(((p + b) / (2 * b) * 128).astype(np.int64), 0, 127)
            m = W.morton3(cell[:, 0], cell[:, 1], cell[:, 2]).astype(np.int64)
            assert np.all(grid[cas, m] > 0)
    assert np.array_equal(W.morton3(np.array([3]), np.array([5]), np.array([7])).astype(np.int32), oracle.morton3D(np.array([[3, 5, 7]])))
    model = W.make_model(0)
    assert model["embeddings"].shape == (6328848, 2) and model["sigma_weights"].shape == (7168,) and model["color_weights"].shape == (11264,)
    assert np.all(model["embeddings"][:4920, 1] == 1.0)                 # the constant-one feature on level 0


def test_get_rays_and_poses():
    from ngp import workload as W
    H = Wd = 16
    intr = W.intrinsics(H, Wd)
    pose = W.orbit_pose(3)
    R = pose[:3, :3]
    assert np.allclose(R.T @ R, np.eye(3), atol=1e-6) and abs(np.linalg.det(R) - 1) < 1e-6
    o, d = W.get_rays(pose, intr, H, Wd)
    assert o.shape == d.shape == (H * Wd, 3) and np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-6)
    # pixel (i, j) -> ((i + 0.5 - cx) / fx, (j + 0.5 - cy) / fy, 1) rotated by the pose (nerf/utils.py:70-108), row-major in j
    j, i = 5, 11
    v = np.array([(i + 0.5 - intr[2]) / intr[0], (j + 0.5 - intr[3]) / intr[1], 1.0])
    v = (v / np.linalg.norm(v)) @ R.T
    assert np.allclose(d[j * Wd + i], v, atol=1e-6)
    assert np.allclose(o, pose[:3, 3])
    centre = d.reshape(H, Wd, 3)[H // 2 - 1:H // 2 + 1, Wd // 2 - 1:Wd // 2 + 1].mean((0, 1))
    assert np.dot(centre / np.linalg.norm(centre), -pose[:3, 3] / np.linalg.norm(pose[:3, 3])) > 0.999   # looks at the origin


def test_sharding_helpers_single_process():
    from ngp import sharding
    for ws in (1, 2, 4, 8):
        views = [sharding.pose_indices(r, ws, 8) for r in range(ws)]
        flat = sorted(v for vs in views for v in vs)
        assert flat == list(range(8 * ws))                              # every view exactly once
        for k in range(8):
            assert len({vs[k] for vs in views}) == ws                   # no two ranks on the same view at a step
        bands = [sharding.row_band(r, ws, 800) for r in range(ws)]
        assert bands[0][0] == 0 and bands[-1][1] == 800 and all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
        assert all((hi - lo) % 8 == 0 for lo, hi in bands[:-1])
    assert sharding.reduce_throughput(10, 2.0, torch.device("cpu")) == (10.0, 2.0)


def test_psnr_formula():
    from oracle import render_oracle as R
    a = np.zeros((4, 3)); b = np.full((4, 3), 0.1)
    assert abs(R.psnr(a, b) - 20.0) < 1e-9                              # -10 log10(0.01) (nerf/utils.py:207)


def test_render_oracle_loop_and_single_march_agree(oracle):
    """the two formulations of the frame (reference loop vs one march per ray) on a tiny view of the S-ring scene"""
    from ngp import workload as W
    from oracle import render_oracle as R
    model = W.make_model(0)
    bf, _ = W.bitfield_from_grid(W.density_grid())
    o, d = W.get_rays(W.orbit_pose(2), W.intrinsics(20, 20), 20, 20)
    f = lambda x, dd: R.field_forward(model, x, dd, 1.0)  # noqa: E731
    tr = []
    a = R.run_cuda(f, o, d, bf, W.BOUND, 2, trace=tr)
    b = R.render_single_march(f, o, d, bf, W.BOUND, 2)
    assert np.max(np.abs(a["image"] - b["image"])) < 1e-6
    assert tr[0][:2] == (400, 1) and all(t[1] == max(min(400 // t[0], 8), 1) for t in tr)       # n_step schedule
    assert b["samples"] <= a["samples"] <= b["marched"].sum()
    assert a["weights_sum"].max() > 0.999 and (a["weights_sum"] == 0).any()                      # saturated and empty rays
