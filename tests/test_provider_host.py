"""CPU: the dataset provider (ngp/provider.py, reference nerf/provider.py:19-337) on datasets written by the test itself.
The reference's provider needs cv2 and trimesh at import (absent here), so its outputs cannot be generated: the expected
values are worked out from its formulas (cited per check) — parity unpinned by reference fixtures."""
import json
import math
import os

import numpy as np
import pytest
import torch

PIL = pytest.importorskip("PIL.Image")

from ngp.nav import get_rays
from ngp.provider import NeRFDataset, nerf_matrix_to_ngp, rand_poses


def c2w(seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    m = np.eye(4)
    m[:3, :3] = q
    m[:3, 3] = rng.uniform(-4, 4, 3)
    return m


def write_images(root, names, H, W, channels, seed=0):
    rng = np.random.default_rng(seed)
    out = {}
    for n in names:
        a = rng.integers(0, 256, size=(H, W, channels), dtype=np.uint8)
        os.makedirs(os.path.dirname(os.path.join(root, n)), exist_ok=True)
        PIL.fromarray(a, "RGBA" if channels == 4 else "RGB").save(os.path.join(root, n))
        out[n] = a
    return out


def test_nerf_matrix_to_ngp_known_answer():
    pose = np.arange(16, dtype=np.float32).reshape(4, 4)
    got = nerf_matrix_to_ngp(pose, scale=0.5, offset=(1, 2, 3))
    # provider.py:21-26: rows 1, 2, 0 of the input; columns 1, 2 negated; t * scale + offset
    want = np.array([[4, -5, -6, 7 * 0.5 + 1], [8, -9, -10, 11 * 0.5 + 2], [0, -1, -2, 3 * 0.5 + 3], [0, 0, 0, 1]], np.float32)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    # a rotation stays a rotation (determinant +1: one cyclic permutation, two sign flips)
    r = nerf_matrix_to_ngp(c2w(3))[:3, :3].astype(np.float64)
    assert abs(np.linalg.det(r) - 1) < 1e-5 and np.allclose(r @ r.T, np.eye(3), atol=1e-5)


def test_rand_poses_look_at_origin():
    g = torch.Generator().manual_seed(0)
    p = rand_poses(64, "cpu", radius=2.5, generator=g)
    centers, forward = p[:, :3, 3], p[:, :3, 2]
    assert torch.allclose(centers.norm(dim=-1), torch.full((64,), 2.5), atol=1e-5)
    assert torch.allclose(forward, -centers / 2.5, atol=1e-5)                   # provider.py:80: forward = -normalize(center)
    r = p[:, :3, :3]
    assert torch.allclose(r @ r.transpose(1, 2), torch.eye(3).expand(64, 3, 3), atol=1e-5)
    theta = torch.acos(centers[:, 1] / 2.5)
    assert theta.min() >= math.pi / 3 - 1e-4 and theta.max() <= 2 * math.pi / 3 + 1e-4


def test_blender_layout_splits_intrinsics_and_collate(tmp_path):
    root = str(tmp_path)
    H, W = 12, 16
    imgs = write_images(root, ["train/r_%d.png" % i for i in range(3)] + ["val/r_0.png"], H, W, 4)
    for split, names in (("train", ["train/r_%d" % i for i in range(3)]), ("val", ["val/r_0"])):
        frames = [{"file_path": "./" + n, "transform_matrix": c2w(hash(n) % 1000).tolist()} for n in names]   # no extension: blender style
        frames.append({"file_path": "./missing/r_9", "transform_matrix": c2w(1).tolist()})                   # skipped (provider.py:207)
        json.dump({"camera_angle_x": 0.6911, "frames": frames}, open(os.path.join(root, "transforms_%s.json" % split), "w"))
    ds = NeRFDataset(root, "cpu", type="train", scale=0.8, offset=(0.1, 0.2, 0.3), num_rays=50)
    assert ds.mode == "blender" and (ds.H, ds.W) == (H, W) and ds.images.shape == (3, H, W, 4) and ds.poses.shape == (3, 4, 4)
    assert np.array_equal((ds.images[1].numpy() * 255).round().astype(np.uint8), imgs["train/r_1.png"])
    fl = W / (2 * np.tan(0.6911 / 2))                                           # provider.py:262-266
    assert np.allclose(ds.intrinsics, [fl, fl, W / 2, H / 2])
    t = json.load(open(os.path.join(root, "transforms_train.json")))["frames"][2]["transform_matrix"]
    assert np.array_equal(ds.poses[2].numpy(), nerf_matrix_to_ngp(np.array(t, np.float32), 0.8, (0.1, 0.2, 0.3)))
    assert abs(ds.radius - ds.poses[:, :3, 3].norm(dim=-1).mean().item()) < 1e-6

    g = torch.Generator().manual_seed(5)
    batch = ds.collate([1], generator=g)
    g = torch.Generator().manual_seed(5)
    rays = get_rays(ds.poses[[1]], ds.intrinsics, H, W, 50, generator=g)
    assert torch.equal(batch["rays_o"], rays["rays_o"]) and torch.equal(batch["rays_d"], rays["rays_d"])
    want = ds.images[1].view(-1, 4)[rays["inds"][0]]
    assert batch["images"].shape == (1, 50, 4) and torch.equal(batch["images"][0], want)

    both = NeRFDataset(root, "cpu", type="trainval")
    assert both.poses.shape[0] == 4 and NeRFDataset(root, "cpu", type="all").poses.shape[0] == 4
    val = NeRFDataset(root, "cpu", type="val")
    out = val.collate([0])
    assert val.num_rays == -1 and out["rays_o"].shape == (1, H * W, 3) and out["images"].shape == (1, H, W, 4)
    assert len(ds.dataloader()) == 3 and ds.dataloader().has_gt


def test_colmap_layout_downscale_error_map_and_test_path(tmp_path):
    root = str(tmp_path)
    H, W = 8, 12
    imgs = write_images(root, ["images/%04d.jpg.png" % i for i in range(4)], H, W, 3, seed=2)
    frames = [{"file_path": "images/%04d.jpg.png" % i, "transform_matrix": c2w(10 + i).tolist()} for i in range(4)]
    json.dump({"fl_x": 20.0, "cx": 6.5, "cy": 3.5, "h": H, "w": W, "frames": frames}, open(os.path.join(root, "transforms.json"), "w"))
    train = NeRFDataset(root, "cpu", type="train", downscale=2, error_map=True, num_rays=16)
    assert train.mode == "colmap" and train.poses.shape[0] == 3 and (train.H, train.W) == (4, 6)     # first frame is the val set
    assert np.allclose(train.intrinsics, [10.0, 10.0, 3.25, 1.75])                                    # provider.py:258-259,271-272
    box = imgs["images/0001.jpg.png"].astype(np.float64).reshape(4, 2, 6, 2, 3).mean(axis=(1, 3)) / 255     # INTER_AREA at factor 2
    assert np.max(np.abs(train.images[0].numpy() - box)) <= 0.5 / 255 + 1e-6
    assert NeRFDataset(root, "cpu", type="val").poses.shape[0] == 1
    batch = train.collate([2], generator=torch.Generator().manual_seed(1))
    assert batch["inds_coarse"].shape == (1, 16) and batch["index"] == [2] and batch["images"].shape == (1, 16, 3)

    test = NeRFDataset(root, "cpu", type="test", n_test=6, rng=np.random.default_rng(0))
    assert test.images is None and test.poses.shape == (7, 4, 4)
    all_poses = np.stack([nerf_matrix_to_ngp(np.array(f["transform_matrix"], np.float32)) for f in frames])
    first, last = test.poses[0].numpy(), test.poses[-1].numpy()
    assert any(np.allclose(first, p, atol=1e-5) for p in all_poses) and any(np.allclose(last, p, atol=1e-5) for p in all_poses)
    mid = test.poses[3].numpy()                                                                       # ratio 0.5: halfway translation
    assert np.allclose(mid[:3, 3], 0.5 * (first[:3, 3] + last[:3, 3]), atol=1e-5)
    assert np.allclose(mid[:3, :3] @ mid[:3, :3].T, np.eye(3), atol=1e-5)
    assert not test.dataloader().has_gt


def test_random_pose_batches(tmp_path):
    root = str(tmp_path)
    write_images(root, ["a.png"], 16, 16, 3)
    json.dump({"camera_angle_x": 0.9, "frames": [{"file_path": "a.png", "transform_matrix": c2w(0).tolist()},
                                                 {"file_path": "a.png", "transform_matrix": c2w(1).tolist()}]},
              open(os.path.join(root, "transforms.json"), "w"))
    ds = NeRFDataset(root, "cpu", type="all", num_rays=64, rand_pose=1)
    assert len(ds.dataloader()) == 2 + 2                                        # provider.py:330-331
    out = ds.collate([3])                                                       # past the end: random pose, low-resolution full image
    assert (out["H"], out["W"]) == (8, 8) and out["rays_o"].shape == (1, 64, 3) and "images" not in out
    with pytest.raises(NotImplementedError):
        NeRFDataset(os.path.join(root, "nope"), "cpu")
