"""GPU, at BASELINE.json's full sizes (800x800 = 640,000 rays, the 6.3 M-row table, 2 x 128^3 grid), where the CPU oracle
is too slow to be the checker: size-independent properties instead.
  * checksums / prefix sums / sortedness of the training march's ray table
  * linearity (composite in rgb; grid encoder in the table)
  * idempotence (packbits of an unpacked bitfield; Morton round trip on the whole lattice)
  * order invariance and repeatability of the fused frame, and agreement of the two GPU render paths with each other"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RES = 800


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def full(dev):
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    model = W.make_model(0)
    grid = W.density_grid()
    field = NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(model)
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
    ren.load_density_grid(grid)
    o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(RES, RES), RES, RES)
    return dict(W=W, ren=ren, o=t(o, dev), d=t(d, dev), grid=grid)


def test_march_train_ray_table_is_a_prefix_sum(full, dev):
    import raymarching
    ren, o, d = full["ren"], full["o"], full["d"]
    N = o.shape[0]
    aabb = ren.aabb_infer
    nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, 2.0, ren.density_bitfield, 2, 128, nears, fars, counter, -1, False,
                                                            128, True, 0.0, 1024)
    r = rays.cpu().numpy().astype(np.int64)
    total = int(counter[0].item())
    assert int(counter[1].item()) == N and total == r[:, 2].sum()                     # checksum of checksums
    assert np.array_equal(r[:, 0], np.arange(N))                                     # slot order = ray order
    assert np.array_equal(r[:, 1], np.concatenate([[0], np.cumsum(r[:-1, 2])]))      # offsets = exclusive prefix sum (sorted)
    assert xyzs.shape[0] == total + (128 - total % 128) and total > 20_000_000
    dl = deltas[:total, 0]
    assert bool((dl > 0).all()) and bool((deltas[total:] == 0).all())                # every reserved slot was filled, padding untouched
    assert bool((xyzs[:total].abs() <= 2.0).all())
    # each ray's directions were copied to each of its samples: the per-ray segment sums recover counts x direction
    seg = torch.repeat_interleave(torch.arange(N, device=dev), rays[:, 2].long())
    acc = torch.zeros(N, 3, device=dev, dtype=torch.float64).index_add_(0, seg, dirs[:total].double())
    want = d.double() * rays[:, 2:3].double()
    assert float((acc - want).abs().max()) < 1e-3


def test_composite_is_linear_in_rgb_and_bounded(full, dev):
    import raymarching
    ren, o, d = full["ren"], full["o"], full["d"]
    nears, fars = raymarching.near_far_from_aabb(o, d, ren.aabb_infer, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, 2.0, ren.density_bitfield, 2, 128, nears, fars, counter, -1, False,
                                                            128, True, 0.0, 1024)
    M = xyzs.shape[0]
    g = torch.Generator(device=dev).manual_seed(0)
    sig = torch.rand(M, device=dev, generator=g) * 30
    rgb = torch.rand(M, 3, device=dev, generator=g)
    ws, dep, img = raymarching.composite_rays_train(sig, rgb, deltas, rays)
    ws2, dep2, img2 = raymarching.composite_rays_train(sig, rgb * 2, deltas, rays)   # scaling by 2 is exact in binary32
    assert torch.equal(img2, img * 2) and torch.equal(ws2, ws) and torch.equal(dep2, dep)
    assert float(ws.max()) <= 1.0 + 1e-5 and float(ws.min()) >= 0.0
    assert bool((img <= ws[:, None] + 1e-5).all())                                    # rgb <= 1 => image <= alpha


def test_grid_encoder_is_linear_in_the_table(full, dev):
    from gridencoder import grid_encode
    W = full["W"]
    offsets, pls = W.grid_offsets(2.0)
    off = t(offsets, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    B = 640128                                                                        # the first march iteration of an 800x800 frame
    x = torch.rand(B, 3, device=dev, generator=g)
    T1 = torch.rand(int(offsets[-1]), 2, device=dev, generator=g) - 0.5
    T2 = torch.rand(int(offsets[-1]), 2, device=dev, generator=g) - 0.5
    e1 = grid_encode(x, T1, off, pls, 16, False, 0, False)
    e2 = grid_encode(x, T2, off, pls, 16, False, 0, False)
    e12 = grid_encode(x, T1 + T2, off, pls, 16, False, 0, False)
    assert e12.shape == (B, 32)
    assert float((e12 - (e1 + e2)).abs().max()) < 4e-6                                # float32 blend of 8 corners
    ez = grid_encode(x, torch.zeros_like(T1), off, pls, 16, False, 0, False)
    assert float(ez.abs().max()) == 0.0


def test_packbits_and_morton_idempotence_full_lattice(full, dev):
    import raymarching
    grid = t(full["grid"], dev)
    bits = raymarching.packbits(grid, 1.0)
    unpacked = torch.from_numpy(np.unpackbits(bits.cpu().numpy(), bitorder="little").astype(np.float32)).to(dev).view(2, -1)
    assert torch.equal(raymarching.packbits(unpacked, 0.5), bits)                     # pack(unpack(pack(g))) == pack(g)
    idx = torch.arange(128 ** 3, dtype=torch.int32, device=dev)
    assert torch.equal(raymarching.morton3D(raymarching.morton3D_invert(idx)), idx)


def test_fused_frame_full_size_properties(full, dev):
    ren, o, d = full["ren"], full["o"], full["d"]
    a = ren.render_fused(o[None], d[None], bg_color=1, image_width=RES)
    b = ren.render_fused(o[None], d[None], bg_color=1, image_width=RES)
    c = ren.render_fused(o[None], d[None], bg_color=1, image_width=0)                 # no tile hint: other traversal order
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["stats"][:3], b["stats"][:3])
    assert torch.equal(a["image"], c["image"]) and torch.equal(a["stats"][:3], c["stats"][:3])   # [3] = tiles: order-dependent
    st = a["stats"].cpu().numpy()
    assert st[1] == 0 and 20_000_000 < st[0] < 40_000_000
    img = a["image"][0]
    assert bool(torch.isfinite(img).all()) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0 + 1e-5
    # a ray with no sample shows exactly the background
    empty = a["weights_sum"] == 0
    assert int(empty.sum()) == RES * RES - int(st[2]) and bool((img[empty] == 1.0).all())
    # the op-by-op loop (reference structure) and the one-launch path agree on the same frame
    tr = []
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        ref = ren.run_cuda(o[None], d[None], bg_color=1, trace=tr)
    assert float((ref["image"][0] - img).abs().max()) < 5e-3
    consumed_upper = sum(k for _, _, k in tr)                                         # marched (>= composited) samples of the loop
    assert st[0] <= consumed_upper <= 1.1 * st[0]


def _constant_density_renderer(dev, grid, density_scale):
    """All-zero networks: sigma = density_scale * exp(0) everywhere, rgb = sigmoid(0).  With a small density_scale no ray
    saturates, so every marched sample is composited and a ray's weights_sum is a strictly increasing function of its
    sample set: one sample more or less anywhere changes it."""
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    field = NGPFieldFF(bound=W.BOUND, density_scale=density_scale).to(dev)
    with torch.no_grad():
        field.sigma_net.weights.zero_()
        field.color_net.weights.zero_()
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_scale=density_scale, density_thresh=0.5).to(dev).eval()
    ren.load_density_grid(grid)
    return ren


def _scene_grid(W, oracle, scene):
    from _util import blob_bitfield
    if scene == "sring":
        return (W.density_grid() > 10.0).astype(np.float32)
    return blob_bitfield(oracle, 2, 128, seed=7, n_blobs=60, bound=W.BOUND)[1]


@pytest.mark.parametrize("scene,dt_gamma,pose", [("sring", 0.0, 1), ("sring", 1.0 / 128, 5), ("blobs", 0.0, 3), ("blobs", 1.0 / 256, 6)])
def test_block_skipping_changes_nothing(full, oracle, dev, scene, dt_gamma, pose):
    """The fused kernel jumps through empty 4^3 / 16^3 blocks of the occupancy grid (render_fused.hip rv_probe).  With the
    switch off it marches cell by cell like the reference.  Both ways, 640,000 rays: bit-identical outputs."""
    import ngp_hip
    W = full["W"]
    ren = _constant_density_renderer(dev, _scene_grid(W, oracle, scene), 1e-3)
    o, d = W.get_rays(W.orbit_pose(pose), W.intrinsics(RES, RES), RES, RES)
    o, d = t(o, dev), t(d, dev)
    L = ngp_hip.lib()
    try:
        assert L.ngp_render_set_block_skip(0) == 1
        plain = ren.render_fused(o[None], d[None], bg_color=1, dt_gamma=dt_gamma, image_width=RES)
        torch.cuda.synchronize()
    finally:
        L.ngp_render_set_block_skip(1)
    skip = ren.render_fused(o[None], d[None], bg_color=1, dt_gamma=dt_gamma, image_width=RES)
    st = skip["stats"].cpu().numpy()
    assert st[1] == 0 and st[0] > 5_000_000 and float(skip["weights_sum"].max()) < 0.5   # nothing saturates: every sample counts
    assert torch.equal(skip["stats"][:3], plain["stats"][:3])
    for key in ("weights_sum", "depth", "image"):
        assert torch.equal(skip[key], plain[key]), f"{key}: {int((skip[key] != plain[key]).sum())} values differ"
    # and the real field (saturating rays, early termination) on the same rays
    a = full["ren"].render_fused(o[None], d[None], bg_color=1, dt_gamma=dt_gamma, image_width=RES)
    try:
        L.ngp_render_set_block_skip(0)
        b = full["ren"].render_fused(o[None], d[None], bg_color=1, dt_gamma=dt_gamma, image_width=RES)
        torch.cuda.synchronize()
    finally:
        L.ngp_render_set_block_skip(1)
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["stats"][:3], b["stats"][:3])


@pytest.mark.parametrize("scene,dt_gamma,pose", [("sring", 0.0, 1), ("sring", 1.0 / 128, 5), ("blobs", 0.0, 3), ("blobs", 1.0 / 256, 6), ("corner", 0.0, 2),
                                                 ("empty", 0.0, 0), ("full", 0.0, 4)])
def test_the_occupied_box_changes_nothing(full, oracle, dev, scene, dt_gamma, pose):
    """A ray of the fused kernel marches no further than where it leaves the box of everything occupied in the grid (render_fused.hip: k_build_coarse's extent,
    the refill's second slab test); beyond that box the reference tests empty cells only.  Switch off = every ray to its own far.  640,000 rays, a constant
    small density (every sample counts, nothing saturates) and the real field: bit-identical images, depths, weights and statistics -- on the ring, on scattered
    blobs, on one small off-centre blob (most rays miss the box), on an empty grid and on a full one (the box is the whole volume)."""
    import ngp_hip
    W = full["W"]
    if scene in ("sring", "blobs"):
        grid = _scene_grid(W, oracle, scene)
    else:
        grid = np.zeros_like(_scene_grid(W, oracle, "sring"))
        if scene == "full":
            grid[:] = 1.0
        elif scene == "corner":
            g = grid.reshape(2, -1)
            idx = oracle.morton3D(np.stack(np.meshgrid(np.arange(100, 110), np.arange(56, 70), np.arange(58, 71), indexing="ij"), -1).reshape(-1, 3).astype(np.int32))
            g[0, idx] = 1.0
    ren = _constant_density_renderer(dev, grid, 1e-3)
    o, d = W.get_rays(W.orbit_pose(pose), W.intrinsics(RES, RES), RES, RES)
    o, d = t(o, dev), t(d, dev)
    L = ngp_hip.lib()
    outs = {}
    for mode in (0, 1):
        previous = L.ngp_render_set_occupied_box(mode)
        try:
            outs[mode] = (ren.render_fused(o[None], d[None], bg_color=1, dt_gamma=dt_gamma, image_width=RES),
                          full["ren"].render_fused(o[None], d[None], bg_color=1, dt_gamma=dt_gamma, image_width=RES) if scene == "sring" else None)
            torch.cuda.synchronize()
        finally:
            L.ngp_render_set_occupied_box(previous)
    for a, b in zip(outs[0], outs[1]):
        if a is None:
            continue
        assert torch.equal(a["stats"][:3], b["stats"][:3])
        for key in ("weights_sum", "depth", "image"):
            assert torch.equal(a[key], b[key]), f"{key}: {int((a[key] != b[key]).sum())} values differ"
    st = outs[1][0]["stats"].cpu().numpy()
    if scene == "empty":
        assert st[0] == 0
    elif scene == "corner":
        assert 0 < st[0] < 3_000_000
    else:
        assert st[0] > 5_000_000


@pytest.mark.parametrize("scene,pose", [("sring", 2), ("blobs", 4)])
def test_fused_march_gives_every_ray_the_single_march_sample_count(full, oracle, dev, scene, pose):
    """Constant density, constant step (dt_gamma 0), no saturation: a ray's weights_sum depends on its sample count alone
    and strictly increases with it.  The counts come from march_rays_train (one march from near to far, bit-exact against
    the cell-by-cell CPU oracle in test_gpu_raymarching.py; it finds empty 16^3 blocks by OR-ing bitfield words, the fused
    kernel through its LDS map): rays with equal counts must have bit-equal weights_sum, more samples more weight --
    i.e. the fused march gave each of the 640,000 rays exactly that many samples."""
    import raymarching
    W = full["W"]
    ren = _constant_density_renderer(dev, _scene_grid(W, oracle, scene), 1e-3)
    o, d = W.get_rays(W.orbit_pose(pose), W.intrinsics(RES, RES), RES, RES)
    o, d = t(o, dev), t(d, dev)
    fused = ren.render_fused(o[None], d[None], bg_color=1, dt_gamma=0, image_width=RES)
    nears, fars = raymarching.near_far_from_aabb(o, d, ren.aabb_infer, ren.min_near)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    _, _, _, rays = raymarching.march_rays_train(o, d, W.BOUND, ren.density_bitfield, ren.cascade, ren.grid_size, nears, fars, counter,
                                                 -1, False, 128, True, 0.0, 1024)
    n = rays[:, 2].long()
    assert int(n.sum()) == int(fused["stats"][0]) and int(n.max()) < 1024
    ws = fused["weights_sum"]
    order = torch.argsort(n)
    n_s, ws_s = n[order], ws[order]
    same = n_s[1:] == n_s[:-1]
    assert bool((ws_s[1:][same] == ws_s[:-1][same]).all()), "two rays with the same sample count differ in weights_sum"
    assert bool((ws_s[1:][~same] > ws_s[:-1][~same]).all()), "weights_sum is not increasing with the sample count"
    assert bool((ws[n == 0] == 0).all())
