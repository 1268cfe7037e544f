"""GPU: the HIP nav queries -- the op chain (ngp.nav.NavQueries over the drop-in gridencoder / shencoder packages) and the fused float32 kernels
(ngp.nav.NativeNavQueries, csrc/nav_field.hip) -- fed through the restated nav/ callers (oracle/nav_oracle.py, pinned on the CPU by
tests/test_nav_golden.py) and compared with what the reference's OWN Planner / Estimator computed (tests/golden/callers_nav.npz):
cost and gradient of Planner.get_state_cost (nav/quad_plot.py:224-254); loss, gradient and the 12 x 12 Hessian of
Estimator.measurement_fn as estimate_state requests it (nav/estimator_helpers.py:293-327,384).  SURVEY 8a N1-N3, 8f-4.

Tolerances (float32 on both sides, different summation orders; the per-query tolerances of tests/test_gpu_nav_native.py propagated):
sigma 2e-5 relative -> cost (sigma^2) 1e-4 relative; planner gradient 2e-3 in norm; filter loss 2e-4 absolute (image 2e-4), gradient 5e-3 in norm,
image term of the Hessian (rotation block) 5e-3 in norm, everything outside that block = the Mahalanobis term to 1e-5.

Also here: composite_rays_train's forward kernel on the operands the executed run() fed its own torch compositing (SURVEY 8c relation 1)."""
import importlib
import os

import numpy as np
import pytest
import torch

importlib.import_module("nerf-navigation_amd")
pytestmark = pytest.mark.gpu

import _nav_cases as NC  # noqa: E402


@pytest.fixture(scope="module")
def queries(dev):
    from ngp import nav
    from ngp import workload as W
    from ngp.field import NGPField
    from ngp.render import NGPRenderer
    g = NC.gold()
    model = W.make_model(0)
    sw, cw = W.nav_weights(0)
    field = NGPField(bound=W.BOUND).to(dev)
    with torch.no_grad():
        field.encoder.embeddings.copy_(torch.from_numpy(model["embeddings"]))
        for layer, w in zip(list(field.sigma_net) + list(field.color_net), sw + cw):
            layer.weight.copy_(torch.from_numpy(w))
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=False).to(dev).eval()
    H, Wd = (int(v) for v in g["mf_HW"])
    kw = dict(num_steps=int(g["mf_num_steps"]))
    return {"chain": nav.NavQueries(ren, g["mf_intrinsics"], H, Wd, **kw), "native": nav.NativeNavQueries(ren, g["mf_intrinsics"], H, Wd, **kw)}


def density_of(queries, which):
    """chain: the level-parallel op chain over the drop-in encoder; native: the one-lane-per-point fused kernels (A*-sized batches); vj: what
    NativeNavQueries.density_fn does with a planner-sized batch -- ONE launch of the value-and-Jacobian kernel, the axis change folded in"""
    if which == "chain":
        return queries["chain"].density_fn
    return queries["native"].density_fn_native if which == "native" else queries["native"].density_fn


@pytest.mark.parametrize("which", ["chain", "native", "vj"])
@pytest.mark.parametrize("tag", ["pl", "plf"])
def test_planner_cost_and_gradient(queries, dev, which, tag):
    g = NC.gold()
    fn = density_of(queries, which)
    with torch.no_grad():
        sigma = fn(torch.from_numpy(g[f"{tag}_points"]).to(dev)).cpu().numpy()
    assert np.max(np.abs(sigma - g[f"{tag}_sigma"]) / g[f"{tag}_sigma"]) < 2e-5
    res = NC.planner_case(g, tag, fn, dev)
    np.testing.assert_allclose(res["per_state"].detach().cpu().numpy(), g[f"{tag}_per_state"], rtol=1e-4)
    np.testing.assert_allclose(res["collision"].detach().cpu().numpy(), g[f"{tag}_collision"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(float(res["total"].detach()), float(g[f"{tag}_total"]), rtol=1e-4)
    assert NC.rel(res["grad_states"].cpu().numpy(), g[f"{tag}_grad_states"]) < 2e-3
    assert NC.rel(res["grad_initial_accel"].cpu().numpy(), g[f"{tag}_grad_initial_accel"]) < 2e-3


def test_planner_through_the_graphed_density(queries, dev):
    """the planner's query as ONE hipGraph replay (ngp.nav.GraphedDensity): the same cost and gradient"""
    from ngp import nav
    g = NC.gold()
    dens = nav.GraphedDensity(queries["chain"], n_points=g["pl_points"].shape[0] * g["pl_points"].shape[1])
    res = NC.planner_case(g, "pl", dens, dev)
    np.testing.assert_allclose(float(res["total"].detach()), float(g["pl_total"]), rtol=1e-4)
    assert NC.rel(res["grad_states"].cpu().numpy(), g["pl_grad_states"]) < 2e-3


@pytest.mark.parametrize("which", ["chain", "native"])
def test_measurement_fn_loss_gradient_and_hessian(queries, dev, which):
    """N3 on the product: torch.autograd.functional.hessian of the measurement function must neither fail nor differentiate through the encoders /
    the fused run() a second time; what it returns is the executed reference's matrix"""
    g = NC.gold()
    q = queries[which]
    res = NC.filter_case(g, q.get_rays_fn, q.render_fn, dev)
    assert abs(float(res["loss"]) - float(g["mf_loss"])) < 2e-4
    assert NC.rel(res["grad"].cpu().numpy(), g["mf_grad"]) < 5e-3
    H = res["hessian"].cpu().numpy()
    image_term = H - g["mf_hessian_process"]
    expected = g["mf_hessian"] - g["mf_hessian_process"]
    outside = np.ones((12, 12), bool); outside[6:9, 6:9] = False
    assert np.abs(image_term[outside]).max() < 1e-5
    assert NC.rel(image_term[6:9, 6:9], expected[6:9, 6:9]) < 5e-3
    assert np.max(np.abs(H - g["mf_hessian"])) < 5e-3 * np.abs(expected).max()


@pytest.mark.parametrize("which", ["chain", "native"])
def test_render_from_pose_view(queries, dev, which):
    from oracle import nav_oracle as NO
    g = NC.gold()
    q = queries[which]
    with torch.no_grad():
        rays = q.get_rays_fn(NO.camera_pose_from_state(torch.from_numpy(g["mf_state"]).to(dev)).reshape(1, 4, 4))
        img = q.render_fn(rays["rays_o"], rays["rays_d"])["image"][0]
    assert np.max(np.abs(img.cpu().numpy() - g["mf_view"])) < 2e-4


def test_hip_compositor_on_the_executed_run_operands(dev):
    """SURVEY 8c relation 1, reference side executed: k_composite_train_fwd on what nerf/renderer.py:206-230 computed (tests/test_nav_golden.py has the
    oracle's side and the tolerances); and bit-equality with the oracle's compositor on the same operands"""
    import raymarching
    from oracle import ngp_oracle as O
    from test_nav_golden import composite_check, composite_inputs
    c = np.load(os.path.join(NC.GOLD, "callers_composite.npz"))
    sigmas, rgbs, deltas, rays = composite_inputs(c)
    t = lambda a: torch.from_numpy(a).to(dev)                                   # noqa: E731
    ws, depth, image = raymarching.composite_rays_train(t(sigmas), t(rgbs), t(deltas), t(rays))
    composite_check(c, ws.cpu().numpy(), image.cpu().numpy())
    ws_o, depth_o, image_o = O.composite_rays_train_forward(sigmas, rgbs, deltas, rays)
    assert np.array_equal(ws.cpu().numpy(), ws_o) and np.array_equal(image.cpu().numpy(), image_o) and np.array_equal(depth.cpu().numpy(), depth_o)


def test_hip_inference_compositor_iterated_on_the_executed_run_operands(dev):
    """k_composite_rays + compact_alive driven like nerf/renderer.py:343-369 over what the executed run() computed: the image within the tolerances of
    tests/test_nav_golden.py, and bit-equal to the oracle's compositor driven the same way"""
    import raymarching
    from oracle import ngp_oracle as O
    from test_nav_golden import composite_check, iterate_composite_rays
    c = np.load(os.path.join(NC.GOLD, "callers_composite.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)             # noqa: E731

    def hip_composite(n_alive, n_step, alive, st, sig, rgb, deltas):
        dev_state = {k: t(v) for k, v in st.items()}
        a = t(alive)
        raymarching.composite_rays(n_alive, n_step, a, dev_state["rays_t"], t(sig), t(rgb), t(deltas), dev_state["ws"], dev_state["depth"], dev_state["image"])
        for k in st:
            st[k][...] = dev_state[k].cpu().numpy()
        return a

    def hip_compact(a):
        packed, cnt = raymarching.compact_alive(a, a.shape[0])
        return packed[: int(cnt.item())].cpu().numpy()

    def cpu_composite(n_alive, n_step, alive, st, sig, rgb, deltas):
        alive = alive.copy()
        O.composite_rays(n_alive, n_step, alive, st["rays_t"], sig, rgb, deltas, st["ws"], st["depth"], st["image"])
        return alive
    ws, image = iterate_composite_rays(c, hip_composite, hip_compact)
    composite_check(c, ws, image)
    ws_o, image_o = iterate_composite_rays(c, cpu_composite, lambda a: a[a >= 0])
    assert np.array_equal(ws, ws_o) and np.array_equal(image, image_o)
