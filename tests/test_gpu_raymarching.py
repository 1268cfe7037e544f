"""GPU parity: every `raymarching` op through the drop-in package (ctypes -> C ABI -> gfx950 kernels) against the
CPU oracle on the same seeded inputs.  Integer/index results and all march/composite float results are BIT-EXACT
(the oracle and the kernels share one arithmetic contract: no FMA contraction, the deterministic exp)."""
import numpy as np
import pytest
import torch

from _util import blob_bitfield, camera_rays

pytestmark = pytest.mark.gpu

BOUND, CAS, H = 2.0, 2, 128
AABB = np.array([-BOUND] * 3 + [BOUND] * 3, np.float32)


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def bits(a):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else a.dtype)


def assert_same_bits(got, want, what):
    g, w = bits(got), bits(want)
    assert g.shape == w.shape, f"{what}: shape {g.shape} vs {w.shape}"
    bad = np.flatnonzero(g.reshape(-1) != w.reshape(-1))
    assert bad.size == 0, f"{what}: {bad.size} of {g.size} words differ, first at {bad[:5]}"


@pytest.fixture(scope="module")
def scene(oracle):
    bitfield, _ = blob_bitfield(oracle, CAS, H, seed=1, bound=BOUND)
    o, d = camera_rays(48, radius=3.2, seed=2)
    nears, fars = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    return dict(bitfield=bitfield, o=o, d=d, nears=nears, fars=fars)


def test_near_far_from_aabb(oracle, dev):
    import raymarching
    o, d = camera_rays(64, radius=3.0, seed=5)
    rng = np.random.default_rng(0)
    o2 = rng.uniform(-4, 4, size=(5000, 3)).astype(np.float32)       # origins inside and outside the box
    d2 = rng.normal(size=(5000, 3)).astype(np.float32)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    for min_near in (0.2, 0.0):
        n_ref, f_ref = oracle.near_far_from_aabb(o, d, AABB, min_near)
        n, f = raymarching.near_far_from_aabb(t(o, dev), t(d, dev), t(AABB, dev), min_near)
        assert_same_bits(n, n_ref, "nears")
        assert_same_bits(f, f_ref, "fars")
    assert (n_ref == np.finfo(np.float32).max).any(), "test must include rays that miss the box"


def test_near_far_accepts_cpu_inputs_like_reference(oracle, dev):
    import raymarching
    o, d = camera_rays(8, seed=1)
    n, f = raymarching.near_far_from_aabb(torch.from_numpy(o), torch.from_numpy(d), t(AABB, dev), 0.2)
    assert n.is_cuda and n.shape == (o.shape[0],)


def test_sph_from_ray(oracle, dev):
    import raymarching
    rng = np.random.default_rng(0)
    o = rng.uniform(-1, 1, size=(4096, 3)).astype(np.float32)
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    got = raymarching.sph_from_ray(t(o, dev), t(d, dev), 4.0).cpu().numpy()
    ref = oracle.sph_from_ray(o, d, 4.0)
    # sqrtf/atan2f come from different libms (ocml vs glibc): tolerance, not bits.  phi wraps at +-1.
    dphi = np.abs(got[:, 1] - ref[:, 1]); dphi = np.minimum(dphi, 2 - dphi)
    assert np.max(np.abs(got[:, 0] - ref[:, 0])) < 2e-6 and dphi.max() < 2e-6


def test_morton_roundtrip_and_oracle(oracle, dev):
    import raymarching
    rng = np.random.default_rng(0)
    c = rng.integers(0, 1024, size=(100000, 3)).astype(np.int32)
    idx = raymarching.morton3D(t(c, dev))
    assert_same_bits(idx, oracle.morton3D(c), "morton3D")
    back = raymarching.morton3D_invert(idx)
    assert_same_bits(back, c, "morton3D_invert(morton3D(c))")
    # the full 128^3 lattice, as update_extra_state sweeps it (nerf/renderer.py:471)
    i = np.arange(128 ** 3, dtype=np.int32)
    assert_same_bits(raymarching.morton3D_invert(t(i, dev)), oracle.morton3D_invert(i), "morton3D_invert lattice")
    assert raymarching.morton3D(t(c[:0], dev)).shape == (0,)


def test_packbits(oracle, dev):
    import raymarching
    rng = np.random.default_rng(0)
    grid = rng.uniform(-1, 30, size=(CAS, H ** 3)).astype(np.float32)
    grid[0, :64] = 10.0                                  # values equal to the threshold are NOT set (strict >)
    grid[1, 100:200] = -1.0                              # untrained cells
    ref = oracle.packbits(grid, 10.0)
    assert np.array_equal(ref, np.packbits(grid.reshape(-1) > 10.0, bitorder="little"))
    got = raymarching.packbits(t(grid, dev), 10.0)
    assert_same_bits(got, ref, "packbits")
    buf = torch.zeros(CAS * H ** 3 // 8, dtype=torch.uint8, device=dev)
    out = raymarching.packbits(t(grid, dev), 10.0, buf)
    assert out.data_ptr() == buf.data_ptr()              # written in place when a bitfield is given
    assert_same_bits(buf, ref, "packbits in place")


@pytest.mark.parametrize("perturb", [0, 7])
@pytest.mark.parametrize("n_step", [1, 4, 8])
@pytest.mark.parametrize("dt_gamma", [0.0, 1.0 / 128])
def test_march_rays_bit_exact(oracle, dev, scene, perturb, n_step, dt_gamma):
    import raymarching
    o, d, nears, fars, bf = scene["o"], scene["d"], scene["nears"], scene["fars"], scene["bitfield"]
    N = o.shape[0]
    rng = np.random.default_rng(3)
    alive = rng.permutation(N)[: N - 37].astype(np.int32)            # a shuffled subset, like a compacted list
    rays_t = nears.copy()
    rays_t[::3] += 0.37                                               # rays resumed mid-way
    x_ref, d_ref, l_ref = oracle.march_rays(alive.size, n_step, alive, rays_t, o, d, BOUND, bf, CAS, H, nears, fars,
                                            128, perturb, dt_gamma, 1024)
    x, dd, l = raymarching.march_rays(alive.size, n_step, t(alive, dev), t(rays_t, dev), t(o, dev), t(d, dev), BOUND,
                                      t(bf, dev), CAS, H, t(nears, dev), t(fars, dev), 128, perturb, dt_gamma, 1024)
    assert x.shape[0] % 128 == 0 and x.shape[0] > alive.size * n_step - 1
    assert_same_bits(l, l_ref, "deltas")
    assert_same_bits(x, x_ref, "xyzs")
    assert_same_bits(dd, d_ref, "dirs")
    assert (l_ref[:, 0] > 0).sum() > 1000, "scene must produce samples"


def test_march_then_composite_loop_bit_exact(oracle, dev, scene):
    """The whole run_cuda inference loop (nerf/renderer.py:343-369) with a synthetic field: alive sets, rays_t and
    the accumulators must match the oracle bit for bit at every iteration."""
    import raymarching
    o, d, nears, fars, bf = scene["o"], scene["d"], scene["nears"], scene["fars"], scene["bitfield"]
    N = o.shape[0]

    def field(xyz):                                                   # a cheap analytic sigma/rgb, float32 on the host
        s = (40.0 * np.exp(-4.0 * (xyz ** 2).sum(1))).astype(np.float32)
        c = (0.5 + 0.5 * np.sin(3.0 * xyz)).astype(np.float32)
        return s, c

    ws_r, dp_r, im_r = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    alive_r, t_r = np.arange(N, dtype=np.int32), nears.copy()
    ws, dp, im = t(ws_r, dev), t(dp_r, dev), t(im_r, dev)
    alive, rt = t(alive_r, dev), t(t_r, dev)
    to, td, tb, tn, tf = t(o, dev), t(d, dev), t(bf, dev), t(nears, dev), t(fars, dev)
    step, iters = 0, 0
    while step < 1024:
        n_alive = alive_r.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        x_r, _, l_r = oracle.march_rays(n_alive, n_step, alive_r, t_r, o, d, BOUND, bf, CAS, H, nears, fars, 128, False, 0.0, 1024)
        s_r, c_r = field(x_r)
        oracle.composite_rays(n_alive, n_step, alive_r, t_r, s_r, c_r, l_r, ws_r, dp_r, im_r)

        x, _, l = raymarching.march_rays(n_alive, n_step, alive, rt, to, td, BOUND, tb, CAS, H, tn, tf, 128, False, 0.0, 1024)
        assert_same_bits(x, x_r, f"xyzs it{iters}")
        raymarching.composite_rays(n_alive, n_step, alive, rt, t(s_r, dev), t(c_r, dev), l, ws, dp, im)
        assert_same_bits(alive, alive_r, f"rays_alive it{iters}")
        assert_same_bits(rt, t_r, f"rays_t it{iters}")

        packed, cnt = raymarching.compact_alive(alive)
        alive_r = alive_r[alive_r >= 0]
        assert int(cnt.item()) == alive_r.shape[0]
        alive = packed[: alive_r.shape[0]].contiguous()
        assert_same_bits(alive, alive_r, f"compacted it{iters}")
        step += n_step
        iters += 1
    assert iters > 5
    assert_same_bits(ws, ws_r, "weights_sum")
    assert_same_bits(dp, dp_r, "depth")
    assert_same_bits(im, im_r, "image")
    assert ws_r.max() > 0.9, "some rays must saturate (exercises the T < 1e-4 exit)"


def test_composite_rays_takes_half_colours_as_the_widened_ones(oracle, dev, scene):
    """a field under autocast returns half colours; the reference's wrapper widens them to float32 (custom_fwd cast_inputs) before its kernel reads them,
    ngp_composite_rays_half widens in the load: the same accumulators and alive list, bit for bit"""
    import raymarching
    o, d, nears, fars, bf = scene["o"], scene["d"], scene["nears"], scene["fars"], scene["bitfield"]
    N = 2000
    assert N <= o.shape[0]
    alive = np.arange(N, dtype=np.int32)
    x, _, l = oracle.march_rays(N, 4, alive, nears.copy(), o, d, BOUND, bf, CAS, H, nears, fars, 128, False, 0.0, 1024)
    rng = np.random.default_rng(1)
    sig = (40.0 * np.exp(-4.0 * (x ** 2).sum(1))).astype(np.float32)
    rgb_h = rng.uniform(0, 1, size=(x.shape[0], 3)).astype(np.float16)
    out = []
    for colours in (t(rgb_h, dev), t(rgb_h.astype(np.float32), dev)):
        a, rt = t(alive, dev), t(nears.copy(), dev)
        ws, dp, im = (torch.zeros(o.shape[0], device=dev), torch.zeros(o.shape[0], device=dev), torch.zeros(o.shape[0], 3, device=dev))
        with torch.autocast("cuda", dtype=torch.float16):
            raymarching.composite_rays(N, 4, a, rt, t(sig, dev), colours, t(l, dev), ws, dp, im)
        out.append((a, rt, ws, dp, im))
    for u, v in zip(*out):
        assert torch.equal(u, v)
    assert float(out[0][2].max()) > 0.2 and int((out[0][0] < 0).sum()) > 0          # four samples per ray: some weight, and rays that ran out of samples


@pytest.mark.parametrize("perturb", [False, True])
@pytest.mark.parametrize("mode", ["first_epoch", "mean_count", "force_all"])
def test_march_rays_train_bit_exact(oracle, dev, scene, perturb, mode):
    import raymarching
    o, d, nears, fars, bf = scene["o"], scene["d"], scene["nears"], scene["fars"], scene["bitfield"]
    kw = dict(perturb=perturb, align=128, dt_gamma=0.0, max_steps=1024)
    if mode == "first_epoch":
        kw.update(mean_count=-1, force_all_rays=False)
    elif mode == "mean_count":
        kw.update(mean_count=20000, force_all_rays=False)            # deliberately too small: later rays are dropped
    else:
        kw.update(mean_count=20000, force_all_rays=True)
    c_ref = np.zeros(2, np.int32)
    x_r, d_r, l_r, r_r = oracle.march_rays_train(o, d, BOUND, bf, CAS, H, nears, fars, c_ref, **kw)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    x, dd, l, r = raymarching.march_rays_train(t(o, dev), t(d, dev), BOUND, t(bf, dev), CAS, H, t(nears, dev), t(fars, dev),
                                               cnt, kw["mean_count"], perturb, 128, kw["force_all_rays"], 0.0, 1024)
    assert_same_bits(cnt, c_ref, "counter")
    assert_same_bits(r, r_r, "rays")
    assert_same_bits(l, l_r, "deltas")
    assert_same_bits(x, x_r, "xyzs")
    assert_same_bits(dd, d_r, "dirs")
    if mode == "mean_count":
        assert (r_r[:, 1] + r_r[:, 2] >= x_r.shape[0]).any(), "test must include dropped rays"


def test_march_rays_train_with_a_pre_advanced_counter(oracle, dev, scene):
    """the public wrapper hands UNINITIALISED buffers to ngp_march_rays_train_filled.  With a caller-supplied step_counter that is not zero
    (point base 777, ray base 0) the slots [0, 777) that no ray fills must read as zeros, like the reference's torch.zeros buffers -- a zero delta is
    what marks a slot as empty (raymarching.cu:548).  The oracle marches into zero-filled buffers."""
    import raymarching
    o, d, nears, fars, bf = scene["o"], scene["d"], scene["nears"], scene["fars"], scene["bitfield"]
    c_ref = np.array([777, 0], np.int32)
    x_r, d_r, l_r, r_r = oracle.march_rays_train(o, d, BOUND, bf, CAS, H, nears, fars, c_ref, mean_count=40000, perturb=False, align=128,
                                                 force_all_rays=False, dt_gamma=0.0, max_steps=1024)
    junk = torch.full((1 << 22,), float("nan"), device=dev)          # poison the allocator's free blocks the wrapper's torch.empty will reuse
    del junk
    cnt = torch.tensor([777, 0], dtype=torch.int32, device=dev)
    x, dd, l, r = raymarching.march_rays_train(t(o, dev), t(d, dev), BOUND, t(bf, dev), CAS, H, t(nears, dev), t(fars, dev), cnt, 40000, False, 128, False, 0.0, 1024)
    assert_same_bits(cnt, c_ref, "counter")
    assert_same_bits(r, r_r, "rays")
    assert_same_bits(l, l_r, "deltas")
    assert_same_bits(x, x_r, "xyzs")
    assert_same_bits(dd, d_r, "dirs")
    assert int(r_r[:, 1].min()) == 777 and not l_r[:777].any()


@pytest.mark.parametrize("max_steps", [1024, 24])
@pytest.mark.parametrize("perturb", [False, True])
def test_wave_per_ray_count_pass_equals_lane_per_ray(oracle, dev, perturb, max_steps):
    """the count pass with one wave per ray (k_march_train_count_wave: 64 lattice points per step, the reference's control flow accepted by one
    ballot per window or replayed run by run) against the one-lane-per-ray pass and against the oracle: counts, order and positions bit for bit -- 4,096 camera rays through the S-ring
    grid plus axis-parallel and missing rays, with and without jitter, and with a max_steps small enough that the cap cuts rays short"""
    import ngp_hip
    import raymarching
    from ngp import workload as W
    from _util import camera_rays
    bf, _ = W.bitfield_from_grid(W.density_grid())
    if max_steps == 24:
        bf = np.full_like(bf, 0xFF)                                     # everything occupied: every lattice point is a sample, long rays hit the cap
    o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(64, 64), 64, 64)
    eo, ed = camera_rays(4, radius=3.0, seed=3)
    o, d = np.concatenate([o, eo[-4:]]), np.concatenate([d, ed[-4:]])
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    c_ref = np.zeros(2, np.int32)
    x_r, _, l_r, r_r = oracle.march_rays_train(o, d, 2.0, bf, 2, 128, nears, fars, c_ref, -1, perturb, 128, False, 0.0, max_steps)
    out = {}
    for wave in (1, 2, 0):                                            # 2: every window through the serial replay (the fallback of 1's one-ballot acceptance)
        ngp_hip.lib().ngp_march_set_wave_per_ray(wave)
        try:
            cnt = torch.zeros(2, dtype=torch.int32, device=dev)
            out[wave] = raymarching.march_rays_train(t(o, dev), t(d, dev), 2.0, t(bf, dev), 2, 128, t(nears, dev), t(fars, dev), cnt, -1, perturb, 128,
                                                     False, 0.0, max_steps) + (cnt,)
        finally:
            ngp_hip.lib().ngp_march_set_wave_per_ray(1)
    for k in range(5):
        assert torch.equal(out[1][k], out[0][k]) and torch.equal(out[2][k], out[0][k]), k
    assert_same_bits(out[1][4], c_ref, "counter")
    assert_same_bits(out[1][3], r_r, "rays")
    assert_same_bits(out[1][0], x_r, "xyzs")
    assert_same_bits(out[1][2], l_r, "deltas")
    if max_steps == 24:
        assert (r_r[:, 2] == max_steps).sum() > 100                       # the cap really cuts rays short


def test_march_rays_train_matches_iterated_march_rays(oracle, scene):
    """SURVEY 8(c) relation 8 (oracle self-consistency, CPU only but kept beside its GPU siblings): marching to
    completion emits the same samples per ray as march_rays_train(force_all_rays, perturb=False)."""
    o, d, nears, fars, bf = scene["o"], scene["d"], scene["nears"], scene["fars"], scene["bitfield"]
    o, d, nears, fars = o[:400], d[:400], nears[:400], fars[:400]
    x_r, _, l_r, rays = oracle.march_rays_train(o, d, BOUND, bf, CAS, H, nears, fars, None, -1, False, -1, True, 0.0, 1024)
    for n in range(0, 400, 13):
        off, cnt = rays[n, 1], rays[n, 2]
        alive = np.array([n], np.int32)
        x1, _, l1 = oracle.march_rays(1, 1024, alive, nears.copy(), o, d, BOUND, bf, CAS, H, nears, fars, -1, False, 0.0, 1024)
        k = int((l1[:, 0] > 0).sum())
        assert k == cnt
        assert np.array_equal(x1[:k].view(np.uint32), x_r[off:off + cnt].view(np.uint32))


@pytest.mark.parametrize("density", [50.0, 150.0], ids=["budget", "every_sample"])
@pytest.mark.parametrize("shape", ["scan", "wave", "lane"])
def test_composite_rays_train_forward_backward(oracle, dev, scene, shape, density):
    """all three kernel shapes (one wave per ray with the chains as lane scans -- the default for training-sized batches --, one wave per ray with every
    lane running the recurrence, one lane per ray) against the oracle, bit for bit.  "budget": perturbed rays under a sample budget that drops some of them;
    "every_sample": rays of up to 525 samples, on a twelfth of which T falls below 1e-4 -- in the first chunk of 64 lanes, the second, or a later one."""
    import ngp_hip
    lib = ngp_hip.lib()
    wave, scan = lib.ngp_march_set_wave_per_ray(shape != "lane"), lib.ngp_composite_set_scan(shape == "scan")
    try:
        _composite_rays_train_forward_backward(oracle, dev, scene, density)
    finally:
        lib.ngp_march_set_wave_per_ray(wave)
        lib.ngp_composite_set_scan(scan)


def _composite_rays_train_forward_backward(oracle, dev, scene, density):
    import raymarching
    o, d, nears, fars, bf = scene["o"], scene["d"], scene["nears"], scene["fars"], scene["bitfield"]
    if density == 50.0:
        x, _, l, rays = oracle.march_rays_train(o, d, BOUND, bf, CAS, H, nears, fars, None, 30000, True, 128, False, 0.0, 1024)
    else:
        x, _, l, rays = oracle.march_rays_train(o, d, BOUND, bf, CAS, H, nears, fars, None, -1, False, 128, True, 0.0, 1024)    # force_all_rays
    rng = np.random.default_rng(0)
    M = x.shape[0]
    sig = (density * np.exp(-3.0 * (x ** 2).sum(1)) * rng.uniform(0.5, 1.5, M)).astype(np.float32)
    rgb = rng.uniform(0, 1, size=(M, 3)).astype(np.float32)
    ws_r, dp_r, im_r = oracle.composite_rays_train_forward(sig, rgb, l, rays)

    ts, tc = t(sig, dev).requires_grad_(True), t(rgb, dev).requires_grad_(True)
    ws, dp, im = raymarching.composite_rays_train(ts, tc, t(l, dev), t(rays, dev))
    assert_same_bits(ws, ws_r, "weights_sum")
    assert_same_bits(dp, dp_r, "depth")
    assert_same_bits(im, im_r, "image")
    if density == 50.0:
        assert (rays[:, 1] + rays[:, 2] >= M).any() and (rays[:, 2] == 0).any()     # dropped and empty rays present
    else:
        assert (rays[:, 2] > 192).sum() > 400 and (ws_r > 0.9999).sum() > 100         # long rays; early exits

    g_ws = rng.normal(size=ws_r.shape).astype(np.float32)
    g_im = rng.normal(size=im_r.shape).astype(np.float32)
    gs_r, gc_r = oracle.composite_rays_train_backward(g_ws, g_im, sig, rgb, l, rays, ws_r, im_r)
    # grad_depth is ignored by the reference (raymarching.py:270): feed a non-zero one and expect no effect
    torch.autograd.backward([ws, dp, im], [t(g_ws, dev), torch.ones_like(dp), t(g_im, dev)])
    assert_same_bits(ts.grad, gs_r, "grad_sigmas")
    assert_same_bits(tc.grad, gc_r, "grad_rgbs")


def test_compact_alive_hands_the_count_to_the_host_without_a_synchronisation(dev):
    """compact_alive(count=True): the kernel stores the count in pinned words the host polls (ngp_compact_alive_publish); the same list and count as the
    synchronising form, call after call (the sequence number tells one call's count from the previous one's)"""
    import raymarching
    rng = np.random.default_rng(3)
    for n in (1, 63, 64, 1000, 70001, 640000, 5):
        a = rng.integers(-1, 50, size=n).astype(np.int32)
        a[rng.random(n) < 0.5] = -1
        ta = t(a, dev)
        packed, cnt, k = raymarching.compact_alive(ta, n, count=True)
        want = a[a >= 0]
        assert k == want.size == int(cnt.item())
        assert np.array_equal(packed[:k].cpu().numpy(), want)
    packed, cnt, k = raymarching.compact_alive(torch.full((300,), -1, dtype=torch.int32, device=dev), 300, count=True)
    assert k == 0


@pytest.mark.parametrize("kind", ["blobs", "empty", "full"])
def test_the_occupied_box_changes_no_sample(oracle, dev, kind):
    """the inference march stops a ray where it leaves the box of everything occupied (formed per call from the coarse map): with the switch on and off, and
    against the oracle (which walks every ray to its far), the same samples (the training march, which does not use the box, beside it) -- a sparse grid, an
    empty and a full one"""
    import ngp_hip
    import raymarching
    bound, cas, grid_h = 2.0, 2, 128
    bf, _ = blob_bitfield(oracle, cas, grid_h, seed=11, n_blobs=3, bound=bound)
    if kind == "empty":
        bf = np.zeros_like(bf)
    elif kind == "full":
        bf = np.full_like(bf, 0xFF)
    o, d = camera_rays(40, radius=3.0, seed=4)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.05)
    N = o.shape[0]
    alive = np.arange(N, dtype=np.int32)
    x_ref, _, l_ref = oracle.march_rays(N, 8, alive, nears.copy(), o, d, bound, bf, cas, grid_h, nears, fars, 128, False, 0.0, 1024)
    ref = oracle.march_rays_train(o, d, bound, bf, cas, grid_h, nears, fars, force_all_rays=True, dt_gamma=0.0, max_steps=256)
    for mode in (1, 0):
        previous = ngp_hip.lib().ngp_march_set_occupied_box(mode)
        try:
            x, _, l = raymarching.march_rays(N, 8, t(alive, dev), t(nears, dev), t(o, dev), t(d, dev), bound, t(bf, dev), cas, grid_h, t(nears, dev), t(fars, dev),
                                             128, False, 0.0, 1024)
            counter = torch.zeros(2, dtype=torch.int32, device=dev)
            xt, _, lt, rays = raymarching.march_rays_train(t(o, dev), t(d, dev), bound, t(bf, dev), cas, grid_h, t(nears, dev), t(fars, dev), counter, -1, False, 128,
                                                           True, 0.0, 256)
        finally:
            ngp_hip.lib().ngp_march_set_occupied_box(previous)
        assert_same_bits(x, x_ref, f"xyzs mode {mode}")
        assert_same_bits(l, l_ref, f"deltas mode {mode}")
        total = int(counter[0].item())
        assert total == int(ref[3][:, 2].sum())
        assert_same_bits(rays, ref[3], f"rays mode {mode}")
        assert_same_bits(xt[:total], ref[0][:total], f"train xyzs mode {mode}")
        assert_same_bits(lt[:total], ref[2][:total], f"train deltas mode {mode}")
    assert (kind == "empty") == (not (l_ref[:, 0] > 0).any())


def test_empty_inputs(dev):
    import raymarching
    z3 = torch.zeros(0, 3, device=dev)
    n, f = raymarching.near_far_from_aabb(z3, z3, t(AABB, dev), 0.2)
    assert n.shape == (0,) and f.shape == (0,)
    out, cnt = raymarching.compact_alive(torch.full((5,), -1, dtype=torch.int32, device=dev))
    assert int(cnt.item()) == 0


@pytest.mark.parametrize("bound,cas,grid_h,dt_gamma", [(1.0, 1, 64, 0.0), (4.0, 3, 64, 1.0 / 256), (2.0, 2, 32, 0.0), (1.5, 2, 128, 0.0),
                                                         (2.0, 2, 128, 1.0 / 64)])
def test_march_to_completion_on_other_grids(oracle, dev, bound, cas, grid_h, dt_gamma):
    """march_rays_train with force_all_rays = every sample of every ray.  Grids on which the kernels skip empty blocks (H >= 64,
    power-of-two bound or one cascade) and grids on which they must not (H = 32; bound 1.5), sparse scenes with long empty
    stretches: sample positions, steps and per-ray counts bit-exact against the oracle, which marches cell by cell."""
    import raymarching
    bf, _ = blob_bitfield(oracle, cas, grid_h, seed=5, n_blobs=6, bound=bound)       # few blobs: mostly empty space
    o, d = camera_rays(40, radius=1.6 * bound, seed=9)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.05)
    ref = oracle.march_rays_train(o, d, bound, bf, cas, grid_h, nears, fars, force_all_rays=True, dt_gamma=dt_gamma, max_steps=1024)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    x, dd, l, rays = raymarching.march_rays_train(t(o, dev), t(d, dev), bound, t(bf, dev), cas, grid_h, t(nears, dev), t(fars, dev),
                                                  counter, -1, False, 128, True, dt_gamma, 1024)
    x_ref, d_ref, l_ref, r_ref = ref[0], ref[1], ref[2], ref[3]
    total = int(counter[0].item())
    assert total == int(r_ref[:, 2].sum()) and total > 2000
    assert_same_bits(rays, r_ref, "rays")
    assert_same_bits(l[:total], l_ref[:total], "deltas")
    assert_same_bits(x[:total], x_ref[:total], "xyzs")


@pytest.mark.parametrize("perturb", [0, 5])
@pytest.mark.parametrize("n_step", [16, 200])
@pytest.mark.parametrize("bound,cas,grid_h", [(1.0, 1, 128), (2.0, 2, 128), (4.0, 3, 64), (1.5, 2, 128)])
def test_march_rays_many_slots_on_sparse_grids_bit_exact(oracle, dev, bound, cas, grid_h, n_step, perturb):
    """march_rays with many slots per ray on sparse scenes (long empty stretches between samples, rays that run out at far before
    their slots are used, resumed rays): positions, directions and both deltas bit-exact against the oracle's cell-by-cell march."""
    import raymarching
    bf, _ = blob_bitfield(oracle, cas, grid_h, seed=11, n_blobs=9, bound=bound)
    o, d = camera_rays(48, radius=1.6 * bound, seed=4)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.05)
    N = o.shape[0]
    alive = np.random.default_rng(2).permutation(N)[: N - 11].astype(np.int32)
    rays_t = nears.copy()
    rays_t[::4] += 0.21 * bound
    x_ref, d_ref, l_ref = oracle.march_rays(alive.size, n_step, alive, rays_t, o, d, bound, bf, cas, grid_h, nears, fars, 128, perturb, 0.0, 1024)
    x, dd, l = raymarching.march_rays(alive.size, n_step, t(alive, dev), t(rays_t, dev), t(o, dev), t(d, dev), bound, t(bf, dev), cas, grid_h,
                                      t(nears, dev), t(fars, dev), 128, perturb, 0.0, 1024)
    per_ray = (l_ref[: alive.size * n_step, 0] > 0).reshape(alive.size, n_step).sum(1)
    assert (per_ray > 6).sum() > 200 and ((per_ray == n_step).any() if n_step == 16 else ((per_ray > 64) & (per_ray < n_step)).any())
    assert_same_bits(l, l_ref, "deltas")
    assert_same_bits(x, x_ref, "xyzs")
    assert_same_bits(dd, d_ref, "dirs")


@pytest.mark.parametrize("n", [1, 255, 256, 257, 70000, 640001])
def test_compact_alive_is_the_stable_mask_at_every_size(dev, n):
    """`rays_alive[rays_alive >= 0]` (nerf/renderer.py:365) as two launches (count, then a write pass whose workgroups add up the counts before them): the same
    elements in the same order and the count, for sizes around the workgroup width, a whole 800 x 800 frame, every-ray-dead, every-ray-alive and sparse survivors"""
    import raymarching
    g = torch.Generator(device="cpu").manual_seed(n)
    for keep in (0.0, 1.0, 0.5, 0.01):
        alive = torch.arange(n, dtype=torch.int32)
        alive[torch.rand(n, generator=g) >= keep] = -1
        packed, cnt = raymarching.compact_alive(alive.to(dev), n)
        want = alive[alive >= 0]
        assert int(cnt.item()) == want.numel()
        assert torch.equal(packed[: want.numel()].cpu(), want)
