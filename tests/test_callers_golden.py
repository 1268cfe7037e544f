"""The oracle's caller restatements (and the product's host-side callers that run without a GPU) against golden vectors produced by
EXECUTING the reference's own Python -- tests/golden/make_callers_golden.py ran nerf/utils.py, nerf/renderer.py and nerf/provider.py
from /root/reference over the oracle's leaf ops.  These pin SURVEY 8a rows R1 (run_cuda inference loop), R2 (training branch), R3 (run +
sample_pdf), R4 (update_extra_state / mark_untrained_grid), R5 (get_rays), R6 (PSNRMeter) and provider.nerf_matrix_to_ngp to executed
reference code.  The leaf kernels under them (march / composite / grid / MLP) remain pinned by the relations of tests/test_oracle_*.py
only: the reference's CUDA cannot be built or run here.

Tolerances: the oracle and the reference run the same torch CPU operations in the same order for everything except where stated, so
most comparisons are bit-exact (`assert_array_equal`)."""
import importlib
import os

import numpy as np
import pytest
import torch

importlib.import_module("nerf-navigation_amd")
from ngp import workload as W  # noqa: E402
from oracle import callers_oracle as CO  # noqa: E402
from oracle import ngp_oracle as O  # noqa: E402
from oracle import render_oracle as RO  # noqa: E402

from _util import ff_model_matrices  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.fixture(scope="module")
def field():
    model = W.make_model(0)
    sw, cw = ff_model_matrices(model)
    return CO.DefaultField(model["embeddings"], model["offsets"], model["per_level_scale"], sw, cw, model["bound"], ff_layout=True)


# ------------------------------------------------------------------------------------------------------------------------
# tier 1: pure functions
# ------------------------------------------------------------------------------------------------------------------------
def test_get_rays_full_image_oracle_and_product():
    """R5, N = -1 (nerf/utils.py:53-116).  The product's torch get_rays runs the reference's operations: bit-exact on the CPU.  The oracle's
    binary32 restatement (and the kernels, csrc/ngp_camera.h) associate the norm and the 3x3 product in a fixed order: <= 2 ulp of 1."""
    from ngp.nav import get_rays
    g = gold("callers_tier1")
    H, Wd = (int(v) for v in g["gr_HW"])
    res = get_rays(torch.from_numpy(g["gr_poses"]), g["gr_intrinsics"], H, Wd)
    np.testing.assert_array_equal(res["rays_o"].numpy(), g["gr_full_o"])
    np.testing.assert_array_equal(res["rays_d"].numpy(), g["gr_full_d"])
    for b in range(2):
        o, d = RO.camera_rays(g["gr_poses"][b], g["gr_intrinsics"], H, Wd)
        np.testing.assert_array_equal(o, g["gr_full_o"][b])
        assert np.max(np.abs(d - g["gr_full_d"][b])) <= 2.5e-7
        o, d = W.get_rays(g["gr_poses"][b], g["gr_intrinsics"], H, Wd)
        assert np.max(np.abs(d - g["gr_full_d"][b])) <= 2.5e-7 and np.array_equal(o, g["gr_full_o"][b])


def test_get_rays_random_and_error_map_branches():
    """R5, N > 0: the same pixels give the same rays; the error-map branch maps the coarse cells and the uniforms it drew to the same
    pixel indices (nerf/utils.py:79-92)."""
    from ngp.nav import get_rays
    g = gold("callers_tier1")
    H, Wd = (int(v) for v in g["gr_HW"])
    poses = torch.from_numpy(g["gr_poses"])
    torch.manual_seed(3)                                               # the product draws from the generator it is given; None = global, like the reference
    res = get_rays(poses, g["gr_intrinsics"], H, Wd, N=100)
    np.testing.assert_array_equal(res["inds"].numpy(), g["gr_rand_inds"])
    np.testing.assert_array_equal(res["rays_d"].numpy(), g["gr_rand_d"])
    np.testing.assert_array_equal(res["rays_o"].numpy(), g["gr_rand_o"])
    for b in range(2):                                                 # the oracle at the same pixels
        o, d = RO.camera_rays(g["gr_poses"][b], g["gr_intrinsics"], H, Wd, inds=g["gr_rand_inds"][b])
        assert np.max(np.abs(d - g["gr_rand_d"][b])) <= 2.5e-7
    torch.manual_seed(4)
    err = torch.rand(2, 128 * 128) + 0.01
    np.testing.assert_array_equal(err.numpy(), g["gr_err_map"])
    res = get_rays(poses, g["gr_intrinsics"], H, Wd, N=64, error_map=err)
    np.testing.assert_array_equal(res["inds_coarse"].numpy(), g["gr_err_inds_coarse"])
    np.testing.assert_array_equal(res["inds"].numpy(), g["gr_err_inds"])
    np.testing.assert_array_equal(res["rays_d"].numpy(), g["gr_err_d"])
    # the index arithmetic of :84-88 restated on the recorded uniforms
    ic, u = g["gr_err_inds_coarse"], g["gr_err_u"]
    sx, sy = H / 128, Wd / 128
    ix = np.minimum((torch.from_numpy(ic // 128) * sx + torch.from_numpy(u[0]) * sx).long().numpy(), H - 1)
    iy = np.minimum((torch.from_numpy(ic % 128) * sy + torch.from_numpy(u[1]) * sy).long().numpy(), Wd - 1)
    np.testing.assert_array_equal(ix * Wd + iy, g["gr_err_inds"])


def test_psnr_meter_oracle_and_product():
    """R6 (nerf/utils.py:185-219)"""
    from ngp.metrics import PSNRMeter
    g = gold("callers_tier1")
    meter = PSNRMeter()
    preds, truths = [], []
    for k in range(3):
        p, t = g[f"psnr_pred{k}"], g[f"psnr_truth{k}"]
        meter.update(torch.from_numpy(p), torch.from_numpy(t))
        preds.append(p); truths.append(t)
        assert meter.measure() == pytest.approx(float(g[f"psnr_after{k}"]), rel=1e-12)
        assert CO.psnr_meter(preds, truths) == pytest.approx(float(g[f"psnr_after{k}"]), rel=1e-6)      # float64 mean vs the reference's float32
    assert meter.report() == str(g["psnr_report"])


def test_sample_pdf_oracle_and_product():
    """R3's sample_pdf (nerf/renderer.py:12-46), det and random (the uniforms it drew are in the file)"""
    from ngp.render import sample_pdf
    g = gold("callers_tier1")
    bins, wts, n = torch.from_numpy(g["pdf_bins"]), torch.from_numpy(g["pdf_weights"]), g["pdf_det"].shape[1]
    np.testing.assert_array_equal(CO.sample_pdf(bins, wts, n, det=True).numpy(), g["pdf_det"])
    np.testing.assert_array_equal(CO.sample_pdf(bins, wts, n, det=False, u=torch.from_numpy(g["pdf_u"])).numpy(), g["pdf_rand"])
    np.testing.assert_array_equal(sample_pdf(bins, wts, n, det=True).numpy(), g["pdf_det"])
    torch.manual_seed(6)
    np.testing.assert_array_equal(sample_pdf(bins, wts, n, det=False).numpy(), g["pdf_rand"])


def test_nerf_matrix_to_ngp_product():
    """provider.py:19-27"""
    from ngp.provider import nerf_matrix_to_ngp
    g = gold("callers_tier1")
    for k, m in enumerate(g["n2n_in"]):
        np.testing.assert_array_equal(nerf_matrix_to_ngp(m), g["n2n_default"][k])
        np.testing.assert_array_equal(nerf_matrix_to_ngp(m, scale=0.8, offset=[0.1, -0.2, 0.3]), g["n2n_scaled"][k])


# ------------------------------------------------------------------------------------------------------------------------
# tier 2: the renderer's control flow over the oracle's leaf ops
# ------------------------------------------------------------------------------------------------------------------------
def _run_case(field, g, tag, training=False):
    ns, us, pert = (int(v) for v in g[f"{tag}_kw"])
    o = torch.from_numpy(g["rays_o"]).clone().requires_grad_(True)
    d = torch.from_numpy(g["rays_d"]).clone().requires_grad_(True)
    res = CO.run(field, o, d, W.BOUND, num_steps=ns, upsample_steps=us, bg_color=1.0, min_near=0.2, training=training,
                 perturb_u=torch.from_numpy(g[f"{tag}_u"]) if pert else None)
    hit = torch.isfinite(res["depth"])
    loss = (res["image"] * torch.from_numpy(g["w_image"])).sum() + (res["depth"][hit] * torch.from_numpy(g["w_depth"])[hit]).sum()
    loss.backward()
    return res, o.grad.numpy(), d.grad.numpy()


@pytest.mark.parametrize("tag", ["fixed", "upsample", "perturb"])
def test_oracle_run_against_reference_run(field, tag):
    """R3 / N2: oracle.callers_oracle.run == NeRFRenderer.run (nerf/renderer.py:125-254) executed: image, depth, weights_sum bit-exact
    (same torch operations in the same order), gradients to the rays to float32 rounding (autograd accumulates in another order)."""
    g = gold("callers_run")
    res, go, gd = _run_case(field, g, tag)
    np.testing.assert_array_equal(res["image"].detach().numpy(), g[f"{tag}_image"])
    np.testing.assert_array_equal(res["weights_sum"].detach().numpy(), g[f"{tag}_weights_sum"])
    np.testing.assert_array_equal(res["depth"].detach().numpy(), g[f"{tag}_depth"])               # NaN for the ray that misses, in both
    for mine, ref in ((go, g[f"{tag}_grad_o"]), (gd, g[f"{tag}_grad_d"])):
        assert np.array_equal(np.isnan(mine), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert np.max(np.abs(mine[ok] - ref[ok])) <= 1e-5 * np.max(np.abs(ref[ok]))


def test_oracle_run_training_mode_resampling(field):
    """run() in training mode resamples with det=False (nerf/renderer.py:187): the uniforms sample_pdf drew are in the file"""
    g = gold("callers_run")
    with torch.no_grad():
        res = CO.run(field, torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"]), W.BOUND, num_steps=48, upsample_steps=32, bg_color=1.0,
                     training=True, pdf_u=torch.from_numpy(g["train_upsample_u"]))
    np.testing.assert_array_equal(res["image"].numpy(), g["train_upsample_image"])
    np.testing.assert_array_equal(res["depth"].numpy(), g["train_upsample_depth"])


@pytest.fixture(scope="module")
def ring_bitfield():
    return W.bitfield_from_grid(W.density_grid())[0]


@pytest.mark.parametrize("tag, dt_gamma, bg", [("inf", 0.0, 1.0), ("inf2", 1 / 128, (0.2, 0.5, 0.7))])
def test_oracle_run_cuda_inference_against_reference_loop(field, ring_bitfield, tag, dt_gamma, bg):
    """R1: oracle.render_oracle.run_cuda == the reference's inference loop (nerf/renderer.py:325-374) executed: per-iteration
    (n_alive, n_step, live samples), image, depth -- bit-exact (same leaf ops, same schedule)."""
    g = gold("callers_run_cuda")

    def fn(x, dd):
        with torch.no_grad():
            s, c = field(torch.from_numpy(x), torch.from_numpy(dd))
        return s.numpy(), c.numpy()
    trace = []
    res = RO.run_cuda(fn, g[f"{tag}_rays_o"], g[f"{tag}_rays_d"], ring_bitfield, W.BOUND, 2, dt_gamma=dt_gamma, bg_color=np.asarray(bg, np.float32),
                      trace=trace)
    np.testing.assert_array_equal(np.array(trace, np.int64), g[f"{tag}_trace"])
    np.testing.assert_array_equal(res["image"], g[f"{tag}_image"])
    np.testing.assert_array_equal(res["depth"], g[f"{tag}_depth"])


def _split_table_digest(g, tag, grad):
    grad = np.asarray(grad, np.float64)
    rows = np.flatnonzero(np.any(grad != 0, axis=1))
    assert rows.size == int(g[f"{tag}_grad_table_n_rows"])
    assert np.linalg.norm(grad) == pytest.approx(float(g[f"{tag}_grad_table_norm"]), rel=1e-5)
    assert np.abs(grad).sum() == pytest.approx(float(g[f"{tag}_grad_table_abs_sum"]), rel=1e-5)
    ref = g[f"{tag}_grad_table_values"]
    assert np.max(np.abs(grad[g[f"{tag}_grad_table_rows"]] - ref)) <= 1e-5 * np.max(np.abs(ref))


@pytest.mark.parametrize("tag, perturb", [("trn", False), ("trnp", True)])
def test_oracle_run_cuda_training_against_reference(field, ring_bitfield, tag, perturb):
    """R2: oracle.callers_oracle.run_cuda_train == the reference's training branch (nerf/renderer.py:282-323) executed: counter, image,
    depth, weights_sum bit-exact; weight and table gradients to float32 rounding."""
    g = gold("callers_run_cuda")
    for p in field.parameters():
        p.grad = None
    counter = np.zeros(2, np.int32)
    res = CO.run_cuda_train(field, g["trn_rays_o"], g["trn_rays_d"], ring_bitfield, W.BOUND, 2, perturb=perturb, counter=counter)
    slot = 0 if tag == "trn" else 1
    np.testing.assert_array_equal(counter, g[f"{tag}_counter"][slot])
    np.testing.assert_array_equal(res["image"].detach().numpy(), g[f"{tag}_image"])
    np.testing.assert_array_equal(res["depth"].detach().numpy(), g[f"{tag}_depth"])
    np.testing.assert_array_equal(res["weights_sum"].detach().numpy(), g[f"{tag}_weights_sum"])
    (res["image"] * torch.from_numpy(g["trn_w_image"])).sum().backward()
    for k, w in enumerate(field.sigma_weights):
        ref = g[f"{tag}_grad_sigma_w{k}"]
        assert np.max(np.abs(w.grad.numpy() - ref)) <= 1e-5 * np.max(np.abs(ref))
    for k, w in enumerate(field.color_weights):
        ref = g[f"{tag}_grad_color_w{k}"]
        assert np.max(np.abs(w.grad.numpy() - ref)) <= 1e-5 * np.max(np.abs(ref))
    _split_table_digest(g, tag, field.embeddings.grad.numpy())


def test_oracle_mean_count_feedback(field, ring_bitfield):
    """the counter ring -> mean_count (nerf/renderer.py:534-537) -> bounded allocation of the next march (raymarching.py:196-203)"""
    g = gold("callers_run_cuda")
    ring = g["trnp_counter"]
    assert int(g["trnp_local_step"]) == 2
    mean_count = int(ring[:2, 0].sum() / 2)
    assert mean_count == int(g["trn3_mean_count"])
    counter = np.zeros(2, np.int32)
    with torch.no_grad():
        res = CO.run_cuda_train(field, g["trn_rays_o"], g["trn_rays_d"], ring_bitfield, W.BOUND, 2, mean_count=mean_count, counter=counter)
    np.testing.assert_array_equal(counter, g["trn3_counter"][2])
    np.testing.assert_array_equal(res["image"].numpy(), g["trn3_image"])


def _density_fn(field):
    def fn(pts):
        with torch.no_grad():
            return field.density(torch.from_numpy(np.ascontiguousarray(pts, np.float32)))["sigma"].numpy()
    return fn


def _duplicate_cells(H, cas, rnd, grid_before):
    """cells of the partial sweep that are written more than once (the reference keeps an arbitrary writer, the oracle the largest)"""
    dup = np.zeros((cas, H ** 3), bool)
    for c in range(cas):
        idx = O.morton3D(np.asarray(rnd["coords"][c], np.int32)).astype(np.int64)
        occ = np.flatnonzero(grid_before[c] > 0)
        if occ.size:
            idx = np.concatenate([idx, occ[np.asarray(rnd["pick"][c], np.int64)]])
        u, n = np.unique(idx, return_counts=True)
        dup[c, u[n > 1]] = True
    return dup


def _bits_equal_outside_gap(bitfield, bitfield_ref, grid_ref, thresh, thresh_ref, exclude, min_share):
    """bits equal on every cell that is not excluded and whose density does not lie between the two thresholds"""
    lo, hi = min(thresh, thresh_ref), max(thresh, thresh_ref)
    decided = ~exclude & ~((grid_ref >= lo) & (grid_ref <= hi))
    bits_ref = np.unpackbits(bitfield_ref, bitorder="little").astype(bool).reshape(grid_ref.shape)
    bits = np.unpackbits(bitfield, bitorder="little").astype(bool).reshape(grid_ref.shape)
    np.testing.assert_array_equal(bits[decided], bits_ref[decided])
    assert decided.mean() > min_share


def test_oracle_update_extra_state_against_reference(field):
    """R4: oracle.callers_oracle.update_extra_state == the reference's (nerf/renderer.py:446-537) executed with the same random numbers:
    full sweep bit-exact (grid, bitfield, mean_density, mean_count); partial sweep bit-exact on every cell written at most once --
    where several samples land in one cell the reference's indexed assignment keeps whichever writer came last and the oracle (like
    the native op) the largest, so there the oracle's value must be >= the reference's."""
    g = gold("callers_grid")
    H, seed, cas = int(g["H"]), int(g["seed"]), 2
    fn = _density_fn(field)
    rnd = CO.grid_update_randoms(seed, 0, cas, H, partial=False)
    grid, bitfield, mean, thresh, _ = CO.update_extra_state(fn, np.zeros((cas, H ** 3), np.float32), W.BOUND, 10.0, 0, rnd, H=H)
    np.testing.assert_array_equal(grid, g["full_grid"])
    # torch.mean sums 65,536 floats in binary32 in its own blocked order (and in another one on a GPU); the oracle and the native op sum in
    # double and round once: equal to a few ulp, and the bitfield equal wherever a cell's density is not inside that gap
    assert mean == pytest.approx(float(g["full_mean_density"]), rel=5e-7)
    _bits_equal_outside_gap(bitfield, g["full_bitfield"], grid, thresh, min(float(g["full_mean_density"]), 10.0), np.zeros_like(grid, bool), 0.999)
    assert int(g["full_mean_count"]) == int((1000 + 1301) / 2)

    n_occ = [(grid[c] > 0).sum() for c in range(cas)]
    rnd = CO.grid_update_randoms(seed, 16, cas, H, partial=True, n_occ=n_occ)
    grid2, bitfield2, mean2, thresh2, _ = CO.update_extra_state(fn, grid, W.BOUND, 10.0, 16, rnd, H=H)
    dup = _duplicate_cells(H, cas, rnd, grid)
    ref = g["partial_grid"]
    assert 0 < dup.sum() < dup.size // 2
    np.testing.assert_array_equal(grid2[~dup], ref[~dup])
    assert np.all(grid2[dup] >= ref[dup])
    # the mean follows its grid: with the reference's choice on the duplicate cells the oracle's mean formula gives the reference's mean
    assert float(np.float32(np.mean(np.clip(np.where(dup, ref, grid2), 0, None), dtype=np.float64))) == pytest.approx(float(g["partial_mean_density"]), rel=5e-7)
    assert mean2 >= float(g["partial_mean_density"]) * (1 - 5e-7)
    _bits_equal_outside_gap(bitfield2, g["partial_bitfield"], ref, thresh2, min(float(g["partial_mean_density"]), 10.0), dup, 0.5)


def test_oracle_mark_untrained_grid_against_reference():
    """R4: mark_untrained_grid (nerf/renderer.py:381-442) executed == the oracle's, cell for cell"""
    g = gold("callers_grid")
    H, cas = int(g["H"]), 2
    grid = CO.mark_untrained_grid(np.zeros((cas, H ** 3), np.float32), g["mark_poses"], g["mark_intrinsics"], W.BOUND, H=H)
    unseen = np.unpackbits(g["mark_unseen"]).astype(bool)[:cas * H ** 3].reshape(cas, -1)
    assert 0 < unseen.sum() < unseen.size
    np.testing.assert_array_equal(grid < 0, unseen)


# ------------------------------------------------------------------------------------------------------------------------
# tier 3: the field models (M1 nerf/network.py, M2 nerf/network_ff.py), executed
# ------------------------------------------------------------------------------------------------------------------------
def _m1_oracle(g):
    model = W.make_model(0)
    w = [g[f"m1_w{k}"] for k in range(7)]
    fld = CO.DefaultField(model["embeddings"], g["m1_offsets"], float(g["m1_per_level_scale"]), w[0:2], w[2:5], W.BOUND)
    shape = tuple(int(v) for v in g["m1_bg_table_shape"])
    bg_table = np.random.default_rng(int(g["m1_bg_table_seed"])).uniform(-1, 1, shape).astype(np.float32)
    return fld, bg_table, w[5:7]


def test_oracle_default_field_against_reference_network():
    """M1: oracle.callers_oracle.DefaultField == NeRFNetwork of nerf/network.py:95-191 executed (encoding.get_encoder, trunc_exp and the layer wiring are the
    reference's; the encoders under it the oracle's): forward, density, color(mask), background and d sigma / d x -- bit-exact except the background's 2-D grid,
    which the oracle evaluates through its C restatement (float32, another summation order: 2e-6)."""
    g = gold("callers_fields")
    fld, bg_table, bg_w = _m1_oracle(g)
    x, d, mask = torch.from_numpy(g["x"]), torch.from_numpy(g["d"]), torch.from_numpy(g["mask"])
    with torch.no_grad():
        sigma, color = fld(x, d)
        dens = fld.density(x)
        cm = fld.color(x, d, mask=mask, **dens)
        bg = fld.background(torch.from_numpy(g["m1_sph"]), d, bg_table, g["m1_bg_offsets"], float(g["m1_bg_per_level_scale"]), bg_w)
    np.testing.assert_array_equal(sigma.numpy(), g["m1_sigma"])
    np.testing.assert_array_equal(color.numpy(), g["m1_color"])
    np.testing.assert_array_equal(dens["sigma"].numpy(), g["m1_density_sigma"])
    np.testing.assert_array_equal(dens["geo_feat"].numpy(), g["m1_geo_feat"])
    np.testing.assert_array_equal(cm.numpy(), g["m1_color_masked"])
    assert np.all(g["m1_color_masked"][~g["mask"]] == 0) and int(g["m1_n_param_groups"]) == 6
    assert np.max(np.abs(bg.numpy() - g["m1_background"])) < 2e-6
    np.testing.assert_array_equal(O.sph_from_ray(g["x"] * 0.1, g["d"], 3.0), g["m1_sph"])


def test_oracle_ff_field_against_reference_network_ff():
    """M2: DefaultField(ff_layout=True) == NeRFNetwork of nerf/network_ff.py:51-134 executed (colour input cat(SH16, geo15, 0) = 32 wide, outputs [:, :3],
    sigma net output 16 wide of which [0] is the logit): bit-exact"""
    g = gold("callers_fields")
    model = W.make_model(0)
    sw, cw = ff_model_matrices(model)
    fld = CO.DefaultField(model["embeddings"], model["offsets"], model["per_level_scale"], sw, cw, model["bound"], ff_layout=True)
    x, d, mask = torch.from_numpy(g["x"]), torch.from_numpy(g["d"]), torch.from_numpy(g["mask"])
    with torch.no_grad():
        sigma, rgb = fld(x, d)
        dens = fld.density(x)
        cm = fld.color(x, d, mask=mask, **dens)
    np.testing.assert_array_equal(sigma.numpy(), g["m2_sigma"])
    np.testing.assert_array_equal(rgb.numpy(), g["m2_rgb"])
    np.testing.assert_array_equal(dens["geo_feat"].numpy(), g["m2_geo_feat"])
    np.testing.assert_array_equal(cm.numpy(), g["m2_color_masked"])
