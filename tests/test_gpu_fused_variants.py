"""GPU: the fused one-launch renderer away from the headline configuration, each case against the CPU oracle's
single-march formulation: bound 1 (one cascade, the Lego-style config 1), dt_gamma > 0 (step grows with t, cascade
chosen by step size), a small max_steps (rays hit the sample cap), rays in arbitrary order / odd counts, and a
non-white per-channel background."""
import numpy as np
import pytest
import torch

from oracle import render_oracle as R

pytestmark = pytest.mark.gpu


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def build(dev, bound, seed=0):
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    model = W.make_model(seed, bound=bound)
    grid = W.density_grid(bound=bound)
    bitfield, _ = W.bitfield_from_grid(grid)
    field = NGPFieldFF(bound=bound).to(dev).load_arrays(model)
    ren = NGPRenderer(field, bound=bound, cuda_ray=True, density_thresh=10.0).to(dev).eval()
    ren.load_density_grid(grid)
    return W, model, bitfield, ren


def check(out, ref, capped_expected=None):
    stats = out["stats"].cpu().numpy()
    img = out["image"].reshape(-1, 3).cpu().numpy()
    assert abs(int(stats[0]) - ref["samples"]) <= max(8, 3e-4 * ref["samples"]), (stats, ref["samples"])
    assert stats[2] == int((ref["consumed"] > 0).sum())
    assert np.max(np.abs(img - ref["image"])) < 5e-3
    assert np.max(np.abs(out["weights_sum"].cpu().numpy() - ref["weights_sum"])) < 5e-3
    if capped_expected is not None:
        assert int(stats[1]) == capped_expected
    return stats


def test_bound_one_single_cascade(oracle, dev):
    W, model, bf, ren = build(dev, 1.0)
    assert ren.cascade == 1 and bf.shape[0] == 128 ** 3 // 8
    o, d = W.get_rays(W.orbit_pose(3, radius=1.3, height=0.5), W.intrinsics(40, 40), 40, 40)
    ref = R.render_single_march(lambda x, dd: R.field_forward(model, x, dd, 1.0), o, d, bf, 1.0, 1)
    out = ren.render_fused(t(o, dev)[None], t(d, dev)[None], bg_color=1, image_width=40)
    check(out, ref, 0)
    assert ref["samples"] > 20000


def test_dt_gamma_and_cascade_from_step(oracle, dev):
    W, model, bf, ren = build(dev, 2.0)
    o, d = W.get_rays(W.orbit_pose(2), W.intrinsics(40, 40), 40, 40)
    for dt_gamma in (1.0 / 128, 1.0 / 32):                      # the reference's default and a coarse one
        ref = R.render_single_march(lambda x, dd: R.field_forward(model, x, dd, 1.0), o, d, bf, 2.0, 2, dt_gamma=dt_gamma)
        out = ren.render_fused(t(o, dev)[None], t(d, dev)[None], dt_gamma=dt_gamma, bg_color=1, image_width=40)
        check(out, ref, 0)


def test_sample_cap_is_counted(oracle, dev):
    W, model, bf, ren = build(dev, 2.0)
    o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(32, 32), 32, 32)
    # every cell occupied: a ray crossing the box takes 4 / dt_min = 1.15 * max_steps steps through mostly thin air
    # (sigma ~ 0.017 keeps T high), so rays that miss the solids run into the sample cap
    max_steps = 64
    bf = np.full_like(bf, 255)
    ren.density_bitfield.fill_(255)
    ref = R.render_single_march(lambda x, dd: R.field_forward(model, x, dd, 1.0), o, d, bf, 2.0, 2, max_steps=max_steps)
    out = ren.render_fused(t(o, dev)[None], t(d, dev)[None], bg_color=1, max_steps=max_steps, image_width=32)
    stats = check(out, ref)
    assert int(ref["marched"].max()) == max_steps              # the oracle's march stopped at the cap for some rays ...
    assert stats[1] > 0                                         # ... and the kernel reports rays that were cut short


def test_unordered_rays_odd_count_and_rgb_background(oracle, dev):
    W, model, bf, ren = build(dev, 2.0)
    o, d = W.get_rays(W.orbit_pose(6), W.intrinsics(48, 48), 48, 48)
    rng = np.random.default_rng(0)
    perm = rng.permutation(o.shape[0])[:2001]                   # shuffled subset, not a multiple of 64
    o, d = o[perm], d[perm]
    bg = (0.2, 0.5, 0.9)
    ref = R.render_single_march(lambda x, dd: R.field_forward(model, x, dd, 1.0), o, d, bf, 2.0, 2, bg_color=np.array(bg, np.float32))
    out = ren.render_fused(t(o, dev)[None], t(d, dev)[None], bg_color=bg, image_width=0)
    check(out, ref, 0)
    # the tile order is a pure permutation of the work: same rays, same results, with or without the hint
    o2, d2 = W.get_rays(W.orbit_pose(6), W.intrinsics(48, 48), 48, 48)
    a = ren.render_fused(t(o2, dev)[None], t(d2, dev)[None], bg_color=1, image_width=48)
    b = ren.render_fused(t(o2, dev)[None], t(d2, dev)[None], bg_color=1, image_width=0)
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["depth"].nan_to_num(), b["depth"].nan_to_num())
    assert torch.equal(a["stats"][:3], b["stats"][:3])


def test_repeatable_bit_for_bit(dev):
    """no atomics on the data path: two launches of the same frame are identical"""
    W, model, bf, ren = build(dev, 2.0)
    o, d = W.get_rays(W.orbit_pose(4), W.intrinsics(64, 64), 64, 64)
    a = ren.render_fused(t(o, dev)[None], t(d, dev)[None], bg_color=1, image_width=64)
    b = ren.render_fused(t(o, dev)[None], t(d, dev)[None], bg_color=1, image_width=64)
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["weights_sum"], b["weights_sum"]) and torch.equal(a["stats"][:3], b["stats"][:3])


def test_three_cascades_without_the_lds_map(oracle, dev):
    """bound 4 = 3 cascades: the coarse occupancy map (12 KiB) no longer fits beside the sample slots, so the frame kernel
    marches through the bitfield itself (no map, no block skipping).  Same answer as the oracle."""
    from _util import blob_bitfield, camera_rays
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    bound = 4.0
    rng = np.random.default_rng(11)
    offsets, pls = W.grid_offsets(bound)
    model = dict(W.make_model(0), bound=bound, offsets=offsets, per_level_scale=pls,
                 embeddings=(rng.uniform(-1, 1, size=(int(offsets[-1]), 2)) * 0.25).astype(np.float32))
    field = NGPFieldFF(bound=bound).to(dev).load_arrays(model)
    ren = NGPRenderer(field, bound=bound, cuda_ray=True, density_thresh=0.5).to(dev).eval()
    assert ren.cascade == 3
    bitfield, grid = blob_bitfield(oracle, 3, 128, seed=4, n_blobs=30, bound=bound)
    ren.load_density_grid(grid)
    assert np.array_equal(ren.density_bitfield.cpu().numpy(), bitfield)
    o, d = camera_rays(24, radius=5.5, seed=3)
    ref = R.render_single_march(lambda x, dd: R.field_forward(model, x, dd, 1.0), o, d, bitfield, bound, 3)
    out = ren.render_fused(t(o, dev)[None], t(d, dev)[None], bg_color=1)
    stats = out["stats"].cpu().numpy()
    assert abs(int(stats[0]) - ref["samples"]) <= max(8, 2e-3 * ref["samples"]) and stats[0] > 1000
    assert np.max(np.abs(out["image"][0].cpu().numpy() - ref["image"])) < 8e-3


@pytest.mark.parametrize("grid_h", [64, 32])
def test_other_grid_sizes(oracle, dev, grid_h):
    """density grids of 64^3 (block skipping still allowed) and 32^3 (coarse map, no skipping) in a sparse scene"""
    from _util import blob_bitfield, camera_rays
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    bound = 2.0
    model = W.make_model(3, bound=bound)
    field = NGPFieldFF(bound=bound).to(dev).load_arrays(model)
    ren = NGPRenderer(field, bound=bound, cuda_ray=True, density_thresh=0.5, grid_size=grid_h).to(dev).eval()
    bitfield, grid = blob_bitfield(oracle, 2, grid_h, seed=8, n_blobs=8, bound=bound)
    ren.load_density_grid(grid)
    assert np.array_equal(ren.density_bitfield.cpu().numpy(), bitfield)
    o, d = camera_rays(32, radius=3.0, seed=6)
    ref = R.render_single_march(lambda x, dd: R.field_forward(model, x, dd, 1.0), o, d, bitfield, bound, 2, H=grid_h)
    out = ren.render_fused(t(o, dev)[None], t(d, dev)[None], bg_color=1)
    stats = out["stats"].cpu().numpy()
    assert abs(int(stats[0]) - ref["samples"]) <= max(8, 2e-3 * ref["samples"]) and stats[0] > 500
    assert stats[2] == int((ref["consumed"] > 0).sum())
    assert np.max(np.abs(out["image"][0].cpu().numpy() - ref["image"])) < 8e-3


@pytest.mark.parametrize("n_rays", [1, 7, 70, 513])
def test_tiny_frames_leave_most_band_queues_empty(oracle, dev, n_rays):
    """fewer 64-ray tiles than ray queues (one per XCD band): empty bands are stepped over, every ray is rendered once"""
    W, model, bf, ren = build(dev, 2.0)
    o, d = W.get_rays(W.orbit_pose(2), W.intrinsics(48, 48), 48, 48)
    pick = np.random.default_rng(n_rays).permutation(o.shape[0])[:n_rays]
    o, d = o[pick], d[pick]
    ref = R.render_single_march(lambda x, dd: R.field_forward(model, x, dd, 1.0), o, d, bf, 2.0, 2)
    out = ren.render_fused(t(o, dev)[None], t(d, dev)[None], bg_color=1)
    stats = out["stats"].cpu().numpy()
    assert abs(int(stats[0]) - ref["samples"]) <= 8 and stats[2] == int((ref["consumed"] > 0).sum())
    assert np.max(np.abs(out["image"].reshape(-1, 3).cpu().numpy() - ref["image"])) < 5e-3
    assert np.isfinite(out["weights_sum"].cpu().numpy()).all()


def test_fused_path_refuses_non_default_fields_and_uses_the_renderers_density_scale(dev):
    """fused_state() packed raw pointers without checking the architecture the kernels are specialised for, and took
    density_scale from the field while the per-op paths use the renderer's (nerf/renderer.py:64 owns the one density_scale)."""
    import numpy as np
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    for kw in (dict(num_layers_color=2), dict(num_layers=3)):
        odd = NGPFieldFF(bound=W.BOUND, **kw).to(dev)
        with pytest.raises(RuntimeError, match="default field"):
            odd.fused_state()
    model = W.make_model(0)
    field = NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(model)
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0, density_scale=0.5).to(dev).eval()
    ren.load_density_grid(W.density_grid())
    o, d = W.get_rays(W.orbit_pose(2), W.intrinsics(48, 48), 48, 48)
    to, td = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
    fused = ren.render_fused(to, td, bg_color=1)["image"]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        perop = ren.run_cuda(to, td, bg_color=1)["image"]
    assert float((fused - perop).abs().max()) < 5e-3
    ren1 = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0, density_scale=1.0).to(dev).eval()
    ren1.load_density_grid(W.density_grid())
    assert float((ren1.render_fused(to, td, bg_color=1)["image"] - fused).abs().max()) > 1e-2      # the scale does reach the kernel
    # moving the module drops the cached half copies (module.to() replaces storage without bumping _version)
    st0 = field.fused_state()
    field.float()
    assert field._fused is None and field.fused_state() is not st0
