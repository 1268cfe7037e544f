"""GPU: the callers of configs 3 (training step) and 4 (nav loop) against oracle/callers_oracle.py on the same seeded inputs.

Rows of SURVEY 8a verified here against the CPU oracle (not against another HIP path): M1 default field, N1 density_fn value +
gradient, N2 / R3 run() image + gradient to the rays, R2 run_cuda training branch image + weight / table gradients, (f4) the nav
drivers (NavQueries, GraphedDensity).  Tolerances are for float32 on both sides with different summation orders; each is stated
where it is used.  The FFMLP field runs under autocast(fp16) and is compared with the float32 oracle holding the same master
weights: its tolerance is the half-precision one."""
import importlib

import numpy as np
import pytest
import torch

importlib.import_module("nerf-navigation_amd")
pytestmark = pytest.mark.gpu

from _util import ff_grads_as_matrices, oracle_field  # noqa: E402


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def linear_model(dev):
    """nn.Linear field (nerf/network.py) with a full-entropy table, the renderer around it and the CPU oracle with the same numbers"""
    from ngp import workload as W
    from ngp.field import NGPField
    from ngp.render import NGPRenderer
    torch.manual_seed(11)
    field = NGPField(bound=W.BOUND).to(dev)
    with torch.no_grad():
        field.encoder.embeddings.uniform_(-0.5, 0.5)
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=False).to(dev).eval()
    return dict(W=W, field=field, ren=ren, oracle=oracle_field(field))


def test_default_field_sigma_rgb_against_oracle(linear_model, dev):
    """M1 (nerf/network.py:95-191): sigma, geo_feat, rgb and color(mask) for fixed weights and points.  float32 both sides:
    32-term and 64-term dot products in different orders -> 2e-5 relative on sigma (exp amplifies the logit error), 2e-6 abs on rgb."""
    field, orc = linear_model["field"], linear_model["oracle"]
    rng = np.random.default_rng(0)
    x = rng.uniform(-2, 2, size=(20000, 3)).astype(np.float32)
    x[:7] = [[2, 2, 2], [-2, -2, -2], [0, 0, 0], [2, -2, 0.5], [1.9999999, 0, 0], [0, 0, -2], [-2, 1, 1]]     # the box faces
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    with torch.no_grad():
        sigma, rgb = field(t(x, dev), t(d, dev))
        dens = field.density(t(x, dev))
        so, co = orc(torch.from_numpy(x), torch.from_numpy(d))
        do = orc.density(torch.from_numpy(x))
    assert np.max(np.abs(sigma.cpu().numpy() - so.numpy()) / so.numpy()) < 2e-5
    assert np.max(np.abs(dens["geo_feat"].cpu().numpy() - do["geo_feat"].numpy())) < 5e-6
    assert np.max(np.abs(rgb.cpu().numpy() - co.numpy())) < 2e-6
    mask = rng.uniform(size=20000) < 0.3
    with torch.no_grad():
        cm = field.color(t(x, dev), t(d, dev), mask=t(mask, dev), **dens)
    assert np.max(np.abs(cm.cpu().numpy()[mask] - co.numpy()[mask])) < 2e-6 and np.all(cm.cpu().numpy()[~mask] == 0)


def test_nav_density_fn_value_and_gradient_against_oracle(linear_model, dev):
    """N1 (simulate.py:340-343, nav/quad_plot.py:224-250): sigma and d sigma / d x on [20,500,3] body points through NavQueries
    (frozen model) and through GraphedDensity (one hipGraph replay).  The gradient is piecewise (trilinear cells): compare in norm
    (1e-4) and per point (1e-3 of the largest gradient)."""
    from ngp import nav
    W, ren, orc = linear_model["W"], linear_model["ren"], linear_model["oracle"]
    q = nav.NavQueries(ren, W.intrinsics(32, 32), 32, 32, freeze=True)
    try:
        rng = np.random.default_rng(1)
        pts = rng.uniform(-1, 1, size=(20, 500, 3)).astype(np.float32)
        w = rng.uniform(0.5, 1.5, size=(20, 500)).astype(np.float32)                 # a weighted sum: every point's gradient is exercised
        pg = t(pts, dev).requires_grad_(True)
        sg = q.density_fn(pg)
        (sg * t(w, dev)).sum().backward()
        po = torch.from_numpy(pts).requires_grad_(True)
        rot = torch.tensor(nav.ROT)
        so = orc.density(po.reshape(-1, 3) @ rot)["sigma"].reshape(20, 500)
        (so * torch.from_numpy(w)).sum().backward()
        assert np.max(np.abs(sg.detach().cpu().numpy() - so.detach().numpy()) / so.detach().numpy()) < 2e-5
        g, go = pg.grad.cpu().numpy(), po.grad.numpy()
        assert rel(g, go) < 1e-4 and np.max(np.abs(g - go)) < 1e-3 * np.abs(go).max()
        dens = nav.GraphedDensity(q, n_points=10000)
        pg2 = t(pts, dev).requires_grad_(True)
        s2 = dens(pg2)
        (s2 * t(w, dev)).sum().backward()
        assert torch.equal(s2.detach(), sg.detach())
        assert rel(pg2.grad.cpu().numpy(), go) < 1e-4
    finally:
        for p in ren.parameters():
            p.requires_grad_(True)


@pytest.mark.parametrize("upsample", [0, 64])
def test_run_image_and_ray_gradients_against_oracle(linear_model, dev, upsample):
    """N2 / R3 (nerf/renderer.py:125-254, simulate.py:346): run() on 1,024 rays x 512 steps (+ 64 sample_pdf steps), staged,
    bg_color 1, perturb False -> image, depth, weights_sum and the gradient of a weighted image sum w.r.t. rays_o and rays_d (what the
    pose filter differentiates, nav/estimator_helpers.py:316).  float32 sums over 512 samples: image 2e-4 abs; gradients 2e-3 in
    norm (5e-3 with resampling) (cell changes of single samples between the two float32 evaluations move individual terms)."""
    from oracle import callers_oracle as CO
    W, ren, orc = linear_model["W"], linear_model["ren"], linear_model["oracle"]
    o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(32, 32), 32, 32)
    rng = np.random.default_rng(2)
    G = rng.uniform(-1, 1, size=(1024, 3)).astype(np.float32)
    ro, rd = t(o, dev)[None].requires_grad_(True), t(d, dev)[None].requires_grad_(True)
    out = ren.render(ro, rd, staged=True, bg_color=1.0, perturb=False, num_steps=512, upsample_steps=upsample, max_ray_batch=4096)
    (out["image"][0] * t(G, dev)).sum().backward()
    co, cd = torch.from_numpy(o).requires_grad_(True), torch.from_numpy(d).requires_grad_(True)
    ref = CO.run(orc, co, cd, W.BOUND, num_steps=512, upsample_steps=upsample, bg_color=1.0)
    (ref["image"] * torch.from_numpy(G)).sum().backward()
    assert np.max(np.abs(out["image"][0].detach().cpu().numpy() - ref["image"].detach().numpy())) < 2e-4
    assert np.max(np.abs(out["depth"][0].detach().cpu().numpy() - ref["depth"].detach().numpy())) < 2e-4
    # with resampling the new depths come from a float32 CDF inversion (differences of nearly equal cumulative sums): 5e-3
    gtol = 5e-3 if upsample else 2e-3
    assert rel(ro.grad[0].cpu().numpy(), co.grad.numpy()) < gtol
    assert rel(rd.grad[0].cpu().numpy(), cd.grad.numpy()) < gtol
    assert float(cd.grad.abs().max()) > 1e-3                                            # a real signal, not 0 == 0


def _training_inputs(W):
    o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(64, 64), 64, 64)                    # 4,096 rays = main_nerf.py's num_rays
    grid = W.density_grid()
    bitfield, _ = W.bitfield_from_grid(grid)
    rng = np.random.default_rng(3)
    target = rng.uniform(0, 1, size=(4096, 3)).astype(np.float32)
    return o, d, grid, bitfield, target


def test_training_step_linear_field_against_oracle(dev):
    """R2 + M1, config 3 in float32 (nn.Linear field, no autocast): one run_cuda training step on 4,096 rays with perturb=True.
    rays / counter / sample positions bit-exact (same pcg32 stream); image 1e-5; d loss / d weights 1e-4 in norm; d loss / d table:
    float atomics sum in any order -> 1e-4 in norm, 1e-3 of the largest entry per element."""
    from ngp import workload as W
    from ngp.field import NGPField
    from ngp.render import NGPRenderer
    from oracle import callers_oracle as CO
    torch.manual_seed(5)
    field = NGPField(bound=W.BOUND).to(dev)
    with torch.no_grad():
        field.encoder.embeddings.uniform_(-0.3, 0.3)
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).train()
    o, d, grid, bitfield, target = _training_inputs(W)
    ren.load_density_grid(grid)
    assert np.array_equal(ren.density_bitfield.cpu().numpy(), bitfield)
    out = ren.run_cuda(t(o, dev)[None], t(d, dev)[None], dt_gamma=0, bg_color=1, perturb=True, force_all_rays=False, max_steps=1024)
    loss = torch.nn.functional.mse_loss(out["image"][0], t(target, dev))
    loss.backward()
    orc = oracle_field(field)
    ref = CO.run_cuda_train(orc, o, d, bitfield, W.BOUND, 2, perturb=True, mean_count=0)
    lo = torch.nn.functional.mse_loss(ref["image"], torch.from_numpy(target))
    lo.backward()
    cnt = ren.step_counter[0].cpu().numpy()
    assert np.array_equal(cnt, ref["counter"]) and cnt[1] == 4096 and cnt[0] > 50000
    assert np.max(np.abs(out["image"][0].detach().cpu().numpy() - ref["image"].detach().numpy())) < 1e-5
    assert abs(float(loss) - float(lo)) < 1e-6 * float(lo)
    for lyr, w in zip(list(field.sigma_net) + list(field.color_net), orc.sigma_weights + orc.color_weights):
        assert rel(lyr.weight.grad.cpu().numpy(), w.grad.numpy()) < 1e-4
    ge, go = field.encoder.embeddings.grad.cpu().numpy(), orc.embeddings.grad.numpy()
    assert rel(ge, go) < 1e-4 and np.max(np.abs(ge - go)) < 1e-3 * np.abs(go).max()
    assert np.count_nonzero(go) > 100000


def test_training_step_ffmlp_autocast_against_oracle(dev):
    """R2 + M2, config 3 as main_nerf.py runs it (--ff --fp16): FFMLP field under autocast with a GradScaler-style loss scale.
    The oracle is float32 with the same master weights, so the tolerance is half precision's: activations and the table are rounded
    to 11 bits, gradients pass through half buffers (ffmlp.cu:749-895, gridencoder.cu:227-343 with at::Half atomics):
    image 4e-3 abs, weight gradients 3e-2 in norm, table gradient 5e-2 in norm."""
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    from oracle import callers_oracle as CO
    torch.manual_seed(6)
    field = NGPFieldFF(bound=W.BOUND).to(dev)
    with torch.no_grad():
        field.encoder.embeddings.uniform_(-0.3, 0.3)
        field.sigma_net.weights.mul_(0.6)                                   # keep sigma in a range where exp() does not saturate halves
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).train()
    o, d, grid, bitfield, target = _training_inputs(W)
    ren.load_density_grid(grid)
    scale = 1024.0
    with torch.autocast("cuda", dtype=torch.float16):
        out = ren.run_cuda(t(o, dev)[None], t(d, dev)[None], dt_gamma=0, bg_color=1, perturb=True, force_all_rays=False, max_steps=1024)
        loss = torch.nn.functional.mse_loss(out["image"][0], t(target, dev))
    (loss * scale).backward()
    orc = oracle_field(field)
    ref = CO.run_cuda_train(orc, o, d, bitfield, W.BOUND, 2, perturb=True, mean_count=0)
    lo = torch.nn.functional.mse_loss(ref["image"], torch.from_numpy(target))
    lo.backward()
    assert np.array_equal(ren.step_counter[0].cpu().numpy(), ref["counter"])
    assert np.max(np.abs(out["image"][0].detach().float().cpu().numpy() - ref["image"].detach().numpy())) < 4e-3
    gs = [g / scale for g in ff_grads_as_matrices(field.sigma_net)]
    gc = [g / scale for g in ff_grads_as_matrices(field.color_net)]
    for k, (g, w) in enumerate(zip(gs + gc, orc.sigma_weights + orc.color_weights)):
        wg = w.grad.numpy()
        if k == len(gs):                                                    # colour input column 31 is the zero pad: its oracle gradient is 0
            assert np.all(wg[:, 31] == 0)
        if k == len(gs) + len(gc) - 1:                                      # colour outputs 3..15 are unused (network_ff.py:72-74)
            assert np.all(wg[3:] == 0) and np.max(np.abs(g[3:])) == 0
        assert rel(g, wg) < 3e-2, k
    ge, go = field.encoder.embeddings.grad.float().cpu().numpy() / scale, orc.embeddings.grad.numpy()
    assert rel(ge, go) < 5e-2


def test_background_model_against_oracle(oracle, dev):
    """bg_radius > 0 (nerf/network.py:69-92,145-161; nerf/renderer.py:233-238,273-278): the background colour of each ray comes from a 2-D hash
    grid over the sphere coordinates (sph_from_ray) and the direction's SH; both renderers mix it in where the rays hit nothing.
    sph_from_ray 2e-6 (two libms), colours 1e-5, images 2e-4."""
    from ngp import workload as W
    from ngp.field import NGPField
    from ngp.render import NGPRenderer
    from oracle import callers_oracle as CO
    torch.manual_seed(9)
    field = NGPField(bound=W.BOUND, bg_radius=3.0).to(dev)
    with torch.no_grad():
        field.encoder.embeddings.uniform_(-0.5, 0.5)
        field.encoder_bg.embeddings.uniform_(-1.0, 1.0)
    assert len(field.get_params(1e-2)) == 6
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=False, bg_radius=3.0).to(dev).eval()
    o, d = W.get_rays(W.orbit_pose(2), W.intrinsics(16, 16), 16, 16)
    to, td = t(o, dev), t(d, dev)
    import raymarching
    sph = raymarching.sph_from_ray(to, td, 3.0)
    sph_ref = oracle.sph_from_ray(o, d, 3.0)
    assert np.max(np.abs(sph.cpu().numpy() - sph_ref)) < 2e-6
    orc = oracle_field(field)
    bgw = [l.weight.detach().cpu().numpy() for l in field.bg_net]
    bg_ref = orc.background(torch.from_numpy(sph_ref), torch.from_numpy(d), field.encoder_bg.embeddings.detach().cpu().numpy(),
                            field.encoder_bg.offsets.cpu().numpy(), float(field.encoder_bg.per_level_scale), bgw)
    with torch.no_grad():
        bg = field.background(sph, td)
        out = ren.run(to[None], td[None], num_steps=64, upsample_steps=0)
    assert np.max(np.abs(bg.cpu().numpy() - bg_ref.numpy())) < 1e-5
    ref = CO.run(orc, torch.from_numpy(o), torch.from_numpy(d), W.BOUND, num_steps=64, upsample_steps=0, bg_color=bg_ref)
    assert np.max(np.abs(out["image"][0].cpu().numpy() - ref["image"].detach().numpy())) < 2e-4
    # the occupancy-grid renderer takes the same route (nerf/renderer.py:273-278)
    ren2 = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, bg_radius=3.0, density_thresh=10.0).to(dev).eval()
    ren2.load_density_grid(W.density_grid())
    with torch.no_grad():
        img = ren2.run_cuda(to[None], td[None])["image"][0]
    assert torch.isfinite(img).all()
    with pytest.raises(ValueError):
        from ngp.field import NGPFieldFF
        NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev), bound=W.BOUND, bg_radius=3.0)
