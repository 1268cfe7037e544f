"""CPU: pins the oracle (oracle/ngp_oracle.c) for the raymarching family.
Known answers: the published PCG32 demo vector and the advance() KATs.  Everything else: the
internal-consistency relations of SURVEY.md 8(c) -- the reference has no fixtures of its own for this path."""
import numpy as np
import torch

from _util import blob_bitfield, camera_rays

BOUND, CAS, H = 2.0, 2, 128
AABB = np.array([-BOUND] * 3 + [BOUND] * 3, np.float32)


def test_pcg32_published_demo_vector(oracle):
    # pcg32-demo output for seed 42, sequence 54 (www.pcg-random.org, pcg32_srandom_r(&rng, 42u, 54u))
    want = [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]
    assert [int(v) for v in oracle.pcg32_stream(42, 54, 6)] == want


def test_pcg32_advance_kats(oracle):
    # pcg32{42}.advance(n).next_uint(), the generator the march kernels use (raymarching.cu:489; pcg32.h:53 seq = 1)
    u, f = oracle.pcg32_kat(42, 1, [0, 1, 2, 63, 4095])
    assert [int(v) for v in u] == [0x4df1ccf9, 0xe5838752, 0x58ed9e10, 0x31b47674, 0x79ffc7eb]
    # advance(n) must equal n single steps, and next_float is the [1,2) mantissa trick
    stream = oracle.pcg32_stream(42, 1, 70)
    assert int(u[3]) == int(stream[63])
    assert np.all((f >= 0) & (f < 1))
    assert f[0] == np.float32(((0x4df1ccf9 >> 9) | 0x3f800000)).view(np.float32) - 1 if False else True
    want = (np.array([(0x4df1ccf9 >> 9) | 0x3f800000], np.uint32).view(np.float32) - np.float32(1))[0]
    assert f[0] == want


def test_expf_is_accurate_and_monotone(oracle):
    x = np.linspace(-87, 10, 200001).astype(np.float32)
    y = oracle.expf(x)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(y - ref) / ref) < 1.2e-7            # ~1 ulp; the reference's __expf is ~2 ulp + the log2e product
    assert np.all(np.diff(y.astype(np.float64)) >= 0)
    assert oracle.expf(np.array([-88.0, -200.0, -np.inf], np.float32)).tolist() == [0.0, 0.0, 0.0]
    assert np.isnan(oracle.expf(np.array([np.nan], np.float32))[0])
    assert oracle.expf(np.array([0.0], np.float32))[0] == 1.0


def test_half_conversions_match_numpy(oracle):
    rng = np.random.default_rng(0)
    a = (rng.normal(size=200000) * np.exp(rng.normal(size=200000) * 8)).astype(np.float32)
    a[:7] = [0.0, -0.0, 65504.0, 65520.0, 6e-8, 2.98e-8, np.inf]
    with np.errstate(over="ignore"):
        assert np.array_equal(oracle.f32_to_f16_bits(a), a.astype(np.float16).view(np.uint16))
    h = np.arange(65536, dtype=np.uint16)
    f = h.view(np.float16).astype(np.float32)
    ok = ~np.isnan(f)
    assert np.array_equal(oracle.f16_bits_to_f32(h).view(np.uint32)[ok], f.view(np.uint32)[ok])


def test_morton_roundtrip_and_packbits(oracle):
    rng = np.random.default_rng(0)
    c = rng.integers(0, 1024, size=(20000, 3)).astype(np.int32)
    idx = oracle.morton3D(c)
    assert np.array_equal(oracle.morton3D_invert(idx), c)
    # bit interleave by definition
    x, y, z = (int(v) for v in c[5])
    want = sum(((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2) for b in range(10))
    assert int(idx[5]) == want
    g = rng.uniform(-1, 20, size=2 * 64 ** 3).astype(np.float32)
    g[:16] = 10.0
    assert np.array_equal(oracle.packbits(g, 10.0), np.packbits(g > 10.0, bitorder="little"))


def test_near_far_slab_test(oracle):
    o = np.array([[0, 0, -3], [0, 0, -3], [5, 5, 5], [0, 0, 0]], np.float32)
    d = np.array([[0, 0, 1], [0.6, 0, 0.8], [1, 0, 0], [0, 0, 1]], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    assert n[0] == 1.0 and f[0] == 5.0
    assert abs(n[1] - 1.25) < 1e-6 and abs(f[1] - 10.0 / 3.0) < 1e-6          # exits through the x = 2 face
    assert n[2] == f[2] == np.finfo(np.float32).max                            # miss
    assert n[3] == np.float32(0.2) and f[3] == 2.0                              # origin inside: near clamps to min_near


def _scene(oracle):
    bf, _ = blob_bitfield(oracle, CAS, H, seed=1, bound=BOUND)
    o, d = camera_rays(24, radius=3.2, seed=2)
    n, f = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    return bf, o, d, n, f


def test_march_train_equals_iterated_march_rays(oracle):
    """relation 8: marching to completion emits the samples of march_rays_train(force_all_rays, perturb=False)"""
    bf, o, d, n, f = _scene(oracle)
    x_r, _, l_r, rays = oracle.march_rays_train(o, d, BOUND, bf, CAS, H, n, f, None, -1, False, -1, True, 0.0, 1024)
    assert np.array_equal(rays[:, 0], np.arange(o.shape[0]))                    # deterministic slot order
    assert np.array_equal(rays[:, 1], np.concatenate([[0], np.cumsum(rays[:-1, 2])]))
    for r in range(0, o.shape[0], 17):
        x1, _, l1 = oracle.march_rays(1, 1024, np.array([r], np.int32), n.copy(), o, d, BOUND, bf, CAS, H, n, f, -1, False, 0.0, 1024)
        k = int((l1[:, 0] > 0).sum())
        assert k == rays[r, 2]
        off = rays[r, 1]
        assert np.array_equal(x1[:k].view(np.uint32), x_r[off:off + k].view(np.uint32))
        assert np.array_equal(l1[:k].view(np.uint32), l_r[off:off + k].view(np.uint32))


def test_march_samples_lie_in_occupied_cells_with_fixed_dt(oracle):
    bf, o, d, n, f = _scene(oracle)
    x, _, l, rays = oracle.march_rays_train(o, d, BOUND, bf, CAS, H, n, f, None, -1, False, -1, True, 0.0, 1024)
    m = int(rays[:, 2].sum())
    assert m > 1000
    dt_min = np.float32(2 * 1.7320508075688772) / np.float32(1024)
    assert np.all(l[:m, 0] == dt_min)                                           # dt_gamma = 0 => every step is dt_min
    assert np.all(l[:m, 1] >= l[:m, 0] - 5e-7)                                  # t - last_t includes skipped space (half an ulp of t)
    # every sample sits in an occupied cell of its cascade level (cell lookup re-derived in numpy)
    mx = np.abs(x[:m]).max(1)
    level = np.clip(np.frexp(mx)[1], 0, CAS - 1)                                # dt*H*0.5 < 0.5 => mip_from_dt = 0
    mb = np.minimum(2.0 ** level, BOUND).astype(np.float32)
    cell = np.clip((0.5 * (x[:m] / mb[:, None] + 1) * H).astype(np.int64), 0, H - 1)
    mort = oracle.morton3D(cell.astype(np.int32)).astype(np.int64) + level * H ** 3
    assert np.all((bf[mort // 8] >> (mort % 8)) & 1)


def test_perturb_shifts_the_start_by_less_than_one_step(oracle):
    bf, o, d, n, f = _scene(oracle)
    x0, _, l0, r0 = oracle.march_rays_train(o, d, BOUND, bf, CAS, H, n, f, None, -1, False, -1, True, 0.0, 1024)
    x1, _, l1, r1 = oracle.march_rays_train(o, d, BOUND, bf, CAS, H, n, f, None, -1, True, -1, True, 0.0, 1024)
    # a sub-step shift changes the count by at most one per occupied interval the ray crosses
    assert np.abs(r0[:, 2] - r1[:, 2]).max() <= 6 and np.abs(r0[:, 2] - r1[:, 2]).mean() < 0.5
    assert not np.array_equal(x0[:100], x1[:100])


def _composite_torch(sig, rgb, deltas):
    """front-to-back compositing as run() writes it (nerf/renderer.py:206-230), float64 torch, differentiable"""
    alpha = 1 - torch.exp(-sig * deltas)
    T = torch.cumprod(torch.cat([torch.ones(1, dtype=sig.dtype), 1 - alpha]), 0)[:-1]
    w = alpha * T
    return w.sum(), (w[:, None] * rgb).sum(0), w, T


def test_composite_train_forward_backward_vs_autograd(oracle):
    """relations 1 and 2: the analytic kernel formulas against autograd of the textbook formulation (no early exit:
    densities kept small so that T stays above 1e-4)"""
    rng = np.random.default_rng(0)
    K = 40
    sig = rng.uniform(0.1, 3.0, K).astype(np.float32)
    rgb = rng.uniform(0, 1, (K, 3)).astype(np.float32)
    dl = np.stack([np.full(K, 0.01, np.float32), np.full(K, 0.01, np.float32)], 1)
    rays = np.array([[0, 0, K]], np.int32)
    pad = np.zeros((8, 1), np.float32)                                           # M must exceed offset+num_steps (:420)
    sig_p = np.concatenate([sig, pad[:, 0]]); rgb_p = np.concatenate([rgb, np.zeros((8, 3), np.float32)])
    dl_p = np.concatenate([dl, np.zeros((8, 2), np.float32)])
    ws, depth, img = oracle.composite_rays_train_forward(sig_p, rgb_p, dl_p, rays)
    ts = torch.tensor(sig, dtype=torch.float64, requires_grad=True)
    tc = torch.tensor(rgb, dtype=torch.float64, requires_grad=True)
    ws_t, img_t, w_t, _ = _composite_torch(ts, tc, torch.tensor(dl[:, 0], dtype=torch.float64))
    assert abs(ws[0] - ws_t.item()) < 1e-6 and np.max(np.abs(img[0] - img_t.detach().numpy())) < 1e-6
    tt = np.cumsum(dl[:, 1].astype(np.float64))
    assert abs(depth[0] - float((w_t.detach().numpy() * tt).sum())) < 1e-6
    g_ws, g_img = np.array([0.7], np.float32), np.array([[0.3, -1.1, 0.5]], np.float32)
    (ws_t * 0.7 + (img_t * torch.tensor(g_img[0], dtype=torch.float64)).sum()).backward()
    gs, gc = oracle.composite_rays_train_backward(g_ws, g_img, sig_p, rgb_p, dl_p, rays, ws, img)
    assert np.max(np.abs(gs[:K] - ts.grad.numpy())) < 2e-6
    assert np.max(np.abs(gc[:K] - tc.grad.numpy())) < 1e-6


def test_composite_early_exit_and_inference_form_agree(oracle):
    """relation 1: composite_rays_train_forward == composite_rays iterated in chunks, except for the documented
    difference in the exit test (T after vs before the sample; 1e-4f vs the double literal 1e-4)"""
    rng = np.random.default_rng(1)
    K = 64
    sig = rng.uniform(20, 80, K).astype(np.float32)                              # opaque quickly
    rgb = rng.uniform(0, 1, (K, 3)).astype(np.float32)
    dl = np.full((K, 2), 0.004, np.float32)
    rays = np.array([[0, 0, K]], np.int32)
    z = np.zeros
    ws_t, d_t, im_t = oracle.composite_rays_train_forward(np.concatenate([sig, z(8, np.float32)]), np.concatenate([rgb, z((8, 3), np.float32)]),
                                                          np.concatenate([dl, z((8, 2), np.float32)]), rays)
    for n_step in (1, 4, 8):
        ws, dp, im = z(1, np.float32), z(1, np.float32), z((1, 3), np.float32)
        alive, t = np.array([0], np.int32), z(1, np.float32)
        k = 0
        while alive[0] >= 0 and k < K:
            oracle.composite_rays(1, n_step, alive, t, sig[k:k + n_step], rgb[k:k + n_step], dl[k:k + n_step], ws, dp, im)
            k += n_step
        # the inference form stops one sample later (it tests T before adding the sample): allow that one sample's weight
        assert 0 <= ws[0] - ws_t[0] < 1.1e-4 and np.max(np.abs(im - im_t)) < 1.1e-4
        assert alive[0] == -1 and ws[0] > 0.9998


def test_composite_rays_marks_finished_rays(oracle):
    ws, dp, im = np.zeros(2, np.float32), np.zeros(2, np.float32), np.zeros((2, 3), np.float32)
    alive, t = np.array([0, 1], np.int32), np.array([0.5, 0.5], np.float32)
    sig = np.array([1.0, 1.0, 1.0, 1.0], np.float32)
    rgb = np.ones((4, 3), np.float32)
    dl = np.array([[0.01, 0.01], [0.01, 0.02], [0.01, 0.01], [0.0, 0.0]], np.float32)   # ray 1 runs out of samples (delta 0)
    oracle.composite_rays(2, 2, alive, t, sig, rgb, dl, ws, dp, im)
    assert alive.tolist() == [0, -1]
    assert t[0] == np.float32(0.5) + np.float32(0.01) + np.float32(0.02) and t[1] == np.float32(0.5)   # dead rays keep their t
    assert ws[0] > ws[1] > 0
