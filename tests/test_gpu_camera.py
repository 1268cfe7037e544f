"""GPU: get_rays as a native op and fused into the frame kernel (reference nerf/utils.py:53-116).
ngp_get_rays is bit-exact against the oracle's binary32 restatement (oracle/render_oracle.py: camera_rays) and equal to
the torch formula of ngp.nav.get_rays to 1e-6; ngp_render_frame_camera is bit-identical to ngp_get_rays + ngp_render_frame."""
import numpy as np
import pytest
import torch

from oracle import render_oracle as R

pytestmark = pytest.mark.gpu


def pose_and_intr(seed, H, W):
    from ngp import workload as Wk
    pose = Wk.orbit_pose(seed).astype(np.float32)
    fx = 0.9 * W
    return pose, (fx, fx * 1.03, W / 2 + 0.37, H / 2 - 1.25)           # off-centre principal point, fx != fy


@pytest.mark.parametrize("H,W", [(40, 56), (1, 1), (7, 130)])
def test_get_rays_full_image_bit_exact(dev, H, W):
    from ngp.nav import get_rays, get_rays_native
    pose, intr = pose_and_intr(3, H, W)
    o, d = get_rays_native(torch.from_numpy(pose), intr, H, W, device=dev)
    ro, rd = R.camera_rays(pose, intr, H, W)
    assert o.shape == (H * W, 3) and np.array_equal(o.cpu().numpy(), ro) and np.array_equal(d.cpu().numpy(), rd)
    t = get_rays(torch.from_numpy(pose)[None].to(dev), intr, H, W)
    assert torch.equal(t["rays_o"][0], o) and (t["rays_d"][0] - d).abs().max().item() < 1e-6
    assert (d.norm(dim=-1) - 1).abs().max().item() < 1e-6


def test_get_rays_selected_pixels(dev):
    from ngp.nav import get_rays_native
    H, W = 33, 47
    pose, intr = pose_and_intr(5, H, W)
    inds = torch.from_numpy(np.random.default_rng(0).integers(0, H * W, size=1000)).to(dev)      # duplicates allowed (:76)
    o, d = get_rays_native(pose.tolist(), intr, H, W, inds=inds)
    full_o, full_d = R.camera_rays(pose, intr, H, W)
    assert np.array_equal(d.cpu().numpy(), full_d[inds.cpu().numpy()]) and np.array_equal(o.cpu().numpy(), full_o[:1000])
    empty = get_rays_native(pose.tolist(), intr, H, W, inds=inds[:0])
    assert empty[0].shape == (0, 3)


def test_get_rays_rejects_bad_arguments(dev):
    import ctypes
    import ngp_hip
    L = ngp_hip.lib()
    pose_h, intr_h = ngp_hip.camera_args(np.eye(4).tolist(), (10.0, 10.0, 4.0, 4.0))
    out = torch.empty(64, 3, device=dev)
    assert L.ngp_get_rays(pose_h, intr_h, 8, 8, None, 63, ngp_hip.ptr(out), ngp_hip.ptr(out), None) != 0      # N != H*W without inds
    assert L.ngp_get_rays(None, intr_h, 8, 8, None, 64, ngp_hip.ptr(out), ngp_hip.ptr(out), None) != 0
    zero_f = (ctypes.c_float * 4)(0.0, 10.0, 4.0, 4.0)
    assert L.ngp_get_rays(pose_h, zero_f, 8, 8, None, 64, ngp_hip.ptr(out), ngp_hip.ptr(out), None) != 0
    with pytest.raises(ValueError):
        ngp_hip.camera_args(np.eye(3).tolist(), (1, 1, 0, 0))


@pytest.mark.parametrize("H,W", [(64, 64), (40, 56), (30, 50)])          # 8x8-tiled and untiled image sizes
def test_frame_from_camera_equals_frame_from_rays(dev, H, W):
    from ngp import workload as Wk
    from ngp.field import NGPFieldFF
    from ngp.nav import get_rays_native
    from ngp.render import NGPRenderer
    model = Wk.make_model(0)
    field = NGPFieldFF(bound=Wk.BOUND).to(dev).load_arrays(model)
    ren = NGPRenderer(field, bound=Wk.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
    ren.load_density_grid(Wk.density_grid())
    pose, intr = pose_and_intr(2, H, W)
    bg = (0.1, 0.6, 0.3)
    a = ren.render_fused_camera(pose, intr, H, W, bg_color=bg)
    o, d = get_rays_native(pose.tolist(), intr, H, W, device=dev)
    b = ren.render_fused(o[None], d[None], bg_color=bg, image_width=W)
    assert a["image"].shape == (H, W, 3) and int(a["stats"][0]) > 1000
    assert torch.equal(a["image"].reshape(-1, 3), b["image"].reshape(-1, 3))
    assert torch.equal(a["depth"].reshape(-1).nan_to_num(), b["depth"].reshape(-1).nan_to_num())
    assert torch.equal(a["weights_sum"], b["weights_sum"]) and torch.equal(a["stats"][:3], b["stats"][:3])
    # and against the oracle's single march on the oracle's rays
    ro, rd = R.camera_rays(pose, intr, H, W)
    bf, _ = Wk.bitfield_from_grid(Wk.density_grid())
    ref = R.render_single_march(lambda x, dd: R.field_forward(model, x, dd, 1.0), ro, rd, bf, Wk.BOUND, 2, bg_color=np.array(bg, np.float32))
    assert np.max(np.abs(a["image"].reshape(-1, 3).cpu().numpy() - ref["image"])) < 5e-3
    assert abs(int(a["stats"][0]) - ref["samples"]) <= max(8, 3e-4 * ref["samples"])


def test_frames_in_one_launch_equal_single_frame_launches(dev):
    """ngp_render_frames_camera: P frames per launch (ramp and drain of the frame kernel paid once): every pixel, depth, weights_sum and the
    summed statistics equal those of P single-frame camera launches, bit for bit; H, W must be multiples of 8; 1..64 frames."""
    from ngp import workload as Wk
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    field = NGPFieldFF(bound=Wk.BOUND).to(dev).load_arrays(Wk.make_model(0))
    ren = NGPRenderer(field, bound=Wk.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
    ren.load_density_grid(Wk.density_grid())
    H, W = 48, 64
    intr = Wk.intrinsics(H, W)
    poses = np.stack([Wk.orbit_pose(k, 5) for k in range(5)])
    bg = (0.2, 0.4, 0.9)
    many = ren.render_fused_cameras(poses, intr, H, W, bg_color=bg)
    assert many["image"].shape == (5, H, W, 3)
    total = torch.zeros(3, dtype=torch.int64, device=dev)
    for k in range(5):
        one = ren.render_fused_camera(poses[k], intr, H, W, bg_color=bg)
        assert torch.equal(many["image"][k], one["image"])
        assert torch.equal(many["depth"][k].nan_to_num(), one["depth"].nan_to_num())
        assert torch.equal(many["weights_sum"][k], one["weights_sum"])
        total += one["stats"][:3].to(torch.int64)
    assert torch.equal(many["stats"][:3].to(torch.int64), total) and int(total[0]) > 10000
    assert torch.equal(ren.render_fused_cameras(poses[:1], intr, H, W, bg_color=bg)["image"][0], many["image"][0])
    with pytest.raises(RuntimeError, match="multiples of 8"):
        ren.render_fused_cameras(poses, Wk.intrinsics(30, 50), 30, 50)
