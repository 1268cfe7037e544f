"""Golden vectors for the caller rows (provider) produced by EXECUTING THE REFERENCE'S OWN PYTHON.

Build container only (needs /root/reference); run as `python -B tests/golden/make_callers_golden.py` from the repository root.
Writes data only: tests/golden/callers_tier1.npz, callers_run.npz, callers_run_cuda.npz, callers_grid.npz, callers_fields.npz.

What executes, and what does not:
  * imported from /root/reference and run on the CPU: nerf/utils.py (get_rays :53-116, PSNRMeter :185-219), nerf/renderer.py
    (sample_pdf :12-46, NeRFRenderer.run :125-254, run_cuda :257-379, mark_untrained_grid :381-442, update_extra_state :446-537),
    nerf/provider.py (nerf_matrix_to_ngp :19-27);
  * the third-party modules those files import but never use on these paths and that this image lacks (cv2, trimesh, imageio,
    tensorboardX, mcubes, torch_ema, lpips) are EMPTY placeholder modules in sys.modules;
  * the reference's extension packages (raymarching, gridencoder, shencoder, ffmlp) are NEVER imported: they are CUDA, importing
    them would hipify into /root/reference.  `raymarching` is a module object whose nine functions are the CPU
    oracle's leaf ops (oracle/ngp_oracle.c, oracle/callers_oracle.py), and the abstract field methods of NeRFRenderer
    (forward / density / color, nerf/renderer.py:103-112) are oracle.callers_oracle.DefaultField.
  So: the CONTROL FLOW AND TENSOR ARITHMETIC OF THE CALLERS is the reference's, executed; the leaf kernels under it are the
  oracle's (they stay "parity unpinned": the reference's CUDA cannot be built here).
  * the field models: nerf/network.py and nerf/network_ff.py are imported and their NeRFNetwork.forward / density / color(mask) /
    background run; `encoding.get_encoder` and `activation.trunc_exp` are the reference's; the modules `gridencoder`, `shencoder`, `ffmlp` they import are
    module objects whose classes (constructor signatures of gridencoder/grid.py:94-106, shencoder/sphere_harmonics.py:62-73, ffmlp/ffmlp.py:100-122) compute
    with the oracle (callers_oracle.grid_encode / sh_encode, a float32 F.linear chain over the reference's flat weight layout);
  * random numbers: get_rays / sample_pdf / run(perturb) draw from torch's global RNG -> the script records what was drawn
    (the product takes the same numbers as inputs).  update_extra_state draws `rand_like` / `randint`: those two functions are
    replaced, for the duration of the call, by readers of the pcg32 streams the native op uses (oracle.callers_oracle
    .grid_update_randoms), so one fixture serves the oracle and the HIP op.
  * nothing is written under /root/reference (python -B / sys.dont_write_bytecode).
The model is ngp.workload.make_model(0) (numpy, seeded), rebuilt by the tests from the same seed: the files hold rays, poses,
random numbers and the reference's outputs only.
"""
import importlib
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

importlib.import_module("nerf-navigation_amd")          # puts ngp/ on the path (workload is numpy only)
from ngp import workload as W  # noqa: E402
from oracle import callers_oracle as CO  # noqa: E402
from oracle import ngp_oracle as O  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import ff_model_matrices  # noqa: E402


# ------------------------------------------------------------------------------------------------------------------------
# placeholders for absent third-party modules, and the `raymarching` module over the oracle's leaf ops
# ------------------------------------------------------------------------------------------------------------------------
def _install_placeholders():
    for name in ("cv2", "trimesh", "imageio", "tensorboardX", "mcubes", "torch_ema", "lpips"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["cv2"].transform = None                               # `from cv2 import transform` (nerf/provider.py:5)
    sys.modules["torch_ema"].ExponentialMovingAverage = object       # `from torch_ema import ExponentialMovingAverage`


def _np(t):
    return t.detach().contiguous().numpy()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


MARCH_LOG = []                                                       # (n_alive, n_step, live samples) per march_rays call


def _oracle_raymarching():
    """module `raymarching` with the wrapper-level signatures of raymarching/raymarching.py over oracle leaf ops"""
    m = types.ModuleType("raymarching")

    def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
        n, f = O.near_far_from_aabb(_np(rays_o), _np(rays_d), _np(aabb), min_near)
        return _t(n), _t(f)

    def sph_from_ray(rays_o, rays_d, radius):
        return _t(O.sph_from_ray(_np(rays_o), _np(rays_d), radius))

    def morton3D(coords):
        return _t(O.morton3D(_np(coords.int())))

    def morton3D_invert(indices):
        return _t(O.morton3D_invert(_np(indices.int())))

    def packbits(grid, thresh, bitfield=None):
        return _t(O.packbits(_np(grid), thresh))

    def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                         align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        counter = step_counter.numpy()                               # a contiguous int32 row of step_counter: updated in place
        assert counter.dtype == np.int32 and counter.flags["C_CONTIGUOUS"]
        x, d, l, r = O.march_rays_train(_np(rays_o), _np(rays_d), bound, _np(density_bitfield), C, H, _np(nears), _np(fars), counter,
                                        mean_count, perturb, align, force_all_rays, dt_gamma, max_steps)
        return _t(x), _t(d), _t(l), _t(r)

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1, perturb=False,
                   dt_gamma=0, max_steps=1024):
        x, d, l = O.march_rays(n_alive, n_step, _np(rays_alive), _np(rays_t), _np(rays_o), _np(rays_d), bound, _np(density_bitfield), C, H,
                               _np(near), _np(far), align, perturb, dt_gamma, max_steps)
        MARCH_LOG.append((int(n_alive), int(n_step), int((l[:, 0] > 0).sum())))
        return _t(x), _t(d), _t(l)

    def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        for a in (rays_alive, rays_t, weights_sum, depth, image):
            assert a.is_contiguous()
        O.composite_rays(n_alive, n_step, rays_alive.numpy(), rays_t.numpy(), _np(sigmas.float()), _np(rgbs.float()), _np(deltas),
                         weights_sum.numpy(), depth.numpy(), image.numpy())            # in place, like the reference's op
        return tuple()

    for fn in (near_far_from_aabb, sph_from_ray, morton3D, morton3D_invert, packbits, march_rays_train, march_rays, composite_rays):
        setattr(m, fn.__name__, fn)
    m.composite_rays_train = CO.composite_rays_train
    return m


ENCODER_RULE = {"first_order": False}          # tier4: differentiate the encoders as the reference's autograd.Functions do (graph-less gradients, N3)


def _oracle_encoder_modules():
    """modules `gridencoder`, `shencoder`, `ffmlp` for nerf/network*.py and encoding.get_encoder: the reference's class names and constructor
    arguments, oracle arithmetic (float32 on the CPU)"""
    import torch.nn as nn
    import torch.nn.functional as F
    g, sh, ff = types.ModuleType("gridencoder"), types.ModuleType("shencoder"), types.ModuleType("ffmlp")

    class GridEncoder(nn.Module):                                    # gridencoder/grid.py:93-156
        def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                     desired_resolution=None, gridtype="hash", align_corners=False):
            super().__init__()
            offsets, pls = O.grid_offsets(input_dim, num_levels, level_dim, per_level_scale, base_resolution, log2_hashmap_size, desired_resolution,
                                          align_corners)
            self.input_dim, self.num_levels, self.level_dim, self.per_level_scale = input_dim, num_levels, level_dim, pls
            self.base_resolution, self.output_dim, self.gridtype_id, self.align_corners = base_resolution, num_levels * level_dim, {"hash": 0, "tiled": 1}[gridtype], align_corners
            self.register_buffer("offsets", torch.from_numpy(offsets))
            self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim).uniform_(-1e-4, 1e-4))

        def forward(self, inputs, bound=1):
            enc = CO.grid_encode_first_order if ENCODER_RULE["first_order"] else CO.grid_encode
            return enc(inputs.view(-1, self.input_dim), self.embeddings, self.offsets.tolist(), self.per_level_scale, self.base_resolution,
                       bound, self.gridtype_id, self.align_corners).view(list(inputs.shape[:-1]) + [self.output_dim])

    class SHEncoder(nn.Module):                                      # shencoder/sphere_harmonics.py:61-87
        def __init__(self, input_dim=3, degree=4):
            super().__init__()
            self.input_dim, self.degree, self.output_dim = input_dim, degree, degree ** 2

        def forward(self, inputs, size=1):
            x = inputs / size
            enc = CO.sh_encode_first_order if ENCODER_RULE["first_order"] else CO.sh_encode
            return enc(x.view(-1, 3), self.degree).view(list(inputs.shape[:-1]) + [self.output_dim])

    class FFMLP(nn.Module):                                          # ffmlp/ffmlp.py:99-168: flat weights [hidden,in] + (n-1) [hidden,hidden] + [16,hidden]
        def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu"):
            super().__init__()
            assert activation == "relu" and output_dim <= 16
            self.input_dim, self.output_dim, self.hidden_dim, self.num_layers, self.padded_output_dim = input_dim, output_dim, hidden_dim, num_layers, 16
            self.weights = nn.Parameter(torch.zeros(hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + 16)))

        def forward(self, inputs):
            shapes = [(self.hidden_dim, self.input_dim)] + [(self.hidden_dim, self.hidden_dim)] * (self.num_layers - 1) + [(16, self.hidden_dim)]
            h, off = inputs, 0
            for k, (r, c) in enumerate(shapes):
                h = F.linear(h, self.weights[off:off + r * c].view(r, c))
                off += r * c
                if k != len(shapes) - 1:
                    h = F.relu(h)
            return h[:, :self.output_dim]                            # ffmlp.py:166: outputs[:B, :self.output_dim]

    g.GridEncoder, sh.SHEncoder, ff.FFMLP = GridEncoder, SHEncoder, FFMLP
    return {"gridencoder": g, "shencoder": sh, "ffmlp": ff}


def import_reference():
    _install_placeholders()
    sys.modules["raymarching"] = _oracle_raymarching()
    sys.modules.update(_oracle_encoder_modules())
    sys.path.insert(0, REF)
    import nerf.provider as P
    import nerf.renderer as R
    import nerf.utils as U
    assert U.__file__.startswith(REF) and R.__file__.startswith(REF) and P.__file__.startswith(REF)
    return U, R, P


def make_renderer(R, field, **kw):
    class OracleFieldRenderer(R.NeRFRenderer):                       # the reference's class; only its abstract methods are filled in
        def forward(self, x, d):
            return field(x, d)

        def density(self, x):
            return field.density(x)

        def color(self, x, d, mask=None, **kwargs):
            return field.color(x, d, mask=mask, **kwargs)

    return OracleFieldRenderer(**kw)


def small_grid(ren, H):
    """the reference hard-codes grid_size = 128 (nerf/renderer.py:74); the grid fixtures use a smaller H so that they stay small"""
    ren.grid_size = H
    ren.density_grid = torch.zeros([ren.cascade, H ** 3])
    ren.density_bitfield = torch.zeros(ren.cascade * H ** 3 // 8, dtype=torch.uint8)
    return ren


class Capture:
    """record / replace the draws of torch.rand, rand_like, randint while the reference code runs"""

    def __init__(self, **replacements):
        self.repl, self.saved, self.log = replacements, {}, []

    def __enter__(self):
        for name in ("rand", "rand_like", "randint", "multinomial"):
            orig = getattr(torch, name)
            self.saved[name] = orig

            def wrapped(*a, _orig=orig, _name=name, **k):
                out = self.repl[_name](*a, **k) if _name in self.repl else _orig(*a, **k)
                self.log.append((_name, out.clone()))
                return out
            setattr(torch, name, wrapped)
        return self

    def __exit__(self, *exc):
        for name, orig in self.saved.items():
            setattr(torch, name, orig)
        return False

    def drawn(self, name):
        return [v for n, v in self.log if n == name]


def oracle_field():
    model = W.make_model(0)
    sw, cw = ff_model_matrices(model)
    return model, CO.DefaultField(model["embeddings"], model["offsets"], model["per_level_scale"], sw, cw, model["bound"], ff_layout=True)


# ------------------------------------------------------------------------------------------------------------------------
def tier1(U, R, P):
    out = {}
    # get_rays: two cameras, a non-square image, off-centre principal point
    poses = np.stack([W.orbit_pose(1), W.orbit_pose(5, radius=2.1, height=-0.3)])
    intr = np.array([41.5, 39.25, 15.75, 12.5], np.float32)
    H, Wd = 24, 32
    out["gr_poses"], out["gr_intrinsics"], out["gr_HW"] = poses, intr, np.array([H, Wd])
    full = U.get_rays(torch.from_numpy(poses), intr, H, Wd)
    out["gr_full_o"], out["gr_full_d"] = _np(full["rays_o"]), _np(full["rays_d"])
    torch.manual_seed(3)
    rnd = U.get_rays(torch.from_numpy(poses), intr, H, Wd, N=100)
    out["gr_rand_inds"], out["gr_rand_o"], out["gr_rand_d"] = _np(rnd["inds"]), _np(rnd["rays_o"]), _np(rnd["rays_d"])
    torch.manual_seed(4)
    err = torch.rand(2, 128 * 128) + 0.01
    with Capture() as cap:
        em = U.get_rays(torch.from_numpy(poses), intr, H, Wd, N=64, error_map=err)
    out["gr_err_map"] = _np(err)
    out["gr_err_inds_coarse"], out["gr_err_inds"] = _np(em["inds_coarse"]), _np(em["inds"])
    out["gr_err_u"] = np.stack([_np(v) for v in cap.drawn("rand")])                 # the two torch.rand(B, N) draws, in order
    out["gr_err_o"], out["gr_err_d"] = _np(em["rays_o"]), _np(em["rays_d"])

    # PSNRMeter: three updates of different shapes
    rng = np.random.default_rng(5)
    meter = U.PSNRMeter()
    for k, shape in enumerate([(1, 50, 3), (1, 8, 8, 3), (2, 30, 3)]):
        t = rng.uniform(0, 1, shape).astype(np.float32)
        p = np.clip(t + rng.normal(scale=0.02 * (k + 1), size=shape), 0, 1).astype(np.float32)
        meter.update(torch.from_numpy(p), torch.from_numpy(t))
        out[f"psnr_pred{k}"], out[f"psnr_truth{k}"] = p, t
        out[f"psnr_after{k}"] = np.float64(meter.measure())
    out["psnr_report"] = np.array(meter.report())

    # sample_pdf: det and random; includes an all-zero weight row and a one-hot row
    B, T, n = 6, 17, 24
    bins = np.sort(rng.uniform(0.2, 3.0, (B, T)).astype(np.float32), axis=1)
    wts = rng.uniform(0, 1, (B, T - 1)).astype(np.float32) ** 4
    wts[1] = 0
    wts[2] = 0; wts[2, 5] = 1
    out["pdf_bins"], out["pdf_weights"] = bins, wts
    out["pdf_det"] = _np(R.sample_pdf(torch.from_numpy(bins), torch.from_numpy(wts), n, det=True))
    torch.manual_seed(6)
    with Capture() as cap:
        out["pdf_rand"] = _np(R.sample_pdf(torch.from_numpy(bins), torch.from_numpy(wts), n, det=False))
    out["pdf_u"] = _np(cap.drawn("rand")[0])

    # nerf_matrix_to_ngp
    mats = rng.normal(size=(4, 4, 4)).astype(np.float64)
    mats[:, 3] = [0, 0, 0, 1]
    out["n2n_in"] = mats
    out["n2n_default"] = np.stack([P.nerf_matrix_to_ngp(m) for m in mats])
    out["n2n_scaled"] = np.stack([P.nerf_matrix_to_ngp(m, scale=0.8, offset=[0.1, -0.2, 0.3]) for m in mats])
    return out


def run_rays():
    o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(12, 12), 12, 12)
    # + an axis-parallel ray, a ray starting inside the box and a ray that misses it
    eo = np.array([[-3, 0.1, 0.2], [0.1, 0.2, 0.3], [5, 5, 5]], np.float32)
    ed = np.array([[1, 0, 0], [0.6, 0.0, -0.8], [1, 0, 0]], np.float32)
    return np.concatenate([o, eo]), np.concatenate([d, ed])


def tier2_run(R):
    """NeRFRenderer.run (nerf/renderer.py:125-254): image / depth / weights_sum and d(sum(image * w) + sum(depth * v)) / d rays"""
    model, field = oracle_field()
    ren = make_renderer(R, field, bound=W.BOUND, cuda_ray=False, min_near=0.2, density_thresh=10).eval()
    ro, rd = run_rays()
    rng = np.random.default_rng(7)
    wi = rng.uniform(0.5, 1.5, (ro.shape[0], 3)).astype(np.float32)
    wd = rng.uniform(0.5, 1.5, ro.shape[0]).astype(np.float32)
    out = {"rays_o": ro, "rays_d": rd, "w_image": wi, "w_depth": wd}
    for tag, kw in (("fixed", dict(num_steps=64, upsample_steps=0)), ("upsample", dict(num_steps=48, upsample_steps=32)),
                    ("perturb", dict(num_steps=64, upsample_steps=0, perturb=True))):
        o = torch.from_numpy(ro)[None].clone().requires_grad_(True)
        d = torch.from_numpy(rd)[None].clone().requires_grad_(True)
        torch.manual_seed(8)
        with Capture() as cap:
            res = ren.run(o, d, bg_color=1.0, **kw)
        hit = torch.isfinite(res["depth"][0])
        loss = (res["image"][0] * torch.from_numpy(wi)).sum() + (res["depth"][0][hit] * torch.from_numpy(wd)[hit]).sum()
        loss.backward()
        out[f"{tag}_image"], out[f"{tag}_depth"], out[f"{tag}_weights_sum"] = _np(res["image"][0]), _np(res["depth"][0]), _np(res["weights_sum"])
        out[f"{tag}_grad_o"], out[f"{tag}_grad_d"] = _np(o.grad[0]), _np(d.grad[0])
        if kw.get("perturb"):
            out[f"{tag}_u"] = _np(cap.drawn("rand")[0])
        out[f"{tag}_kw"] = np.array([kw["num_steps"], kw["upsample_steps"], int(bool(kw.get("perturb")))])
    # training-mode resampling (det=False: sample_pdf draws torch.rand)
    ren.train()
    torch.manual_seed(9)
    with Capture() as cap, torch.no_grad():
        res = ren.run(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], bg_color=1.0, num_steps=48, upsample_steps=32)
    out["train_upsample_image"], out["train_upsample_depth"] = _np(res["image"][0]), _np(res["depth"][0])
    out["train_upsample_u"] = _np(cap.drawn("rand")[0])
    return out


def table_grad_digest(g, n_keep=4096, seed=0):
    """a 50 MB table gradient as (rows touched, l2 norm, sum, |.|-sum, values at n_keep seeded touched rows)"""
    g = np.asarray(g, np.float64)
    rows = np.flatnonzero(np.any(g != 0, axis=1))
    pick = np.sort(np.random.default_rng(seed).choice(rows, size=min(n_keep, rows.size), replace=False))
    return dict(n_rows=np.int64(rows.size), norm=np.float64(np.linalg.norm(g)), sum=np.float64(g.sum()), abs_sum=np.float64(np.abs(g).sum()),
                rows=pick.astype(np.int64), values=g[pick].astype(np.float32))


def tier2_run_cuda(R):
    """NeRFRenderer.run_cuda (nerf/renderer.py:257-379), both branches, on the S-ring scene's analytic occupancy grid"""
    model, field = oracle_field()
    ren = make_renderer(R, field, bound=W.BOUND, cuda_ray=True, min_near=0.2, density_thresh=10)
    grid = W.density_grid()
    bitfield, _ = W.bitfield_from_grid(grid)
    ren.density_grid.copy_(torch.from_numpy(grid))
    ren.density_bitfield.copy_(torch.from_numpy(bitfield))
    out = {}
    # ---- inference branch: the alive-count schedule, compaction, background mix, depth normalisation
    ro, rd = W.get_rays(W.orbit_pose(1), W.intrinsics(20, 20), 20, 20)
    ro, rd = np.concatenate([ro, [[5, 5, 5]]]).astype(np.float32), np.concatenate([rd, [[1, 0, 0]]]).astype(np.float32)      # + a miss: depth 0/0
    ren.eval()
    del MARCH_LOG[:]
    with torch.no_grad():
        res = ren.run_cuda(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], dt_gamma=0, bg_color=1.0, perturb=False, max_steps=1024)
    out["inf_rays_o"], out["inf_rays_d"] = ro, rd
    out["inf_trace"] = np.array(MARCH_LOG, np.int64)
    out["inf_image"], out["inf_depth"] = _np(res["image"][0]), _np(res["depth"][0])
    # a second view with dt_gamma > 0 and a coloured background
    ro2, rd2 = W.get_rays(W.orbit_pose(4), W.intrinsics(16, 16), 16, 16)
    del MARCH_LOG[:]
    with torch.no_grad():
        res = ren.run_cuda(torch.from_numpy(ro2)[None], torch.from_numpy(rd2)[None], dt_gamma=1 / 128, bg_color=torch.tensor([0.2, 0.5, 0.7]),
                           perturb=False, max_steps=1024)
    out["inf2_rays_o"], out["inf2_rays_d"] = ro2, rd2
    out["inf2_trace"] = np.array(MARCH_LOG, np.int64)
    out["inf2_image"], out["inf2_depth"] = _np(res["image"][0]), _np(res["depth"][0])

    # ---- training branch: counter ring, mean_count feedback, image, gradients of sum(image * w) to every parameter
    ren.train()
    ro, rd = W.get_rays(W.orbit_pose(2), W.intrinsics(16, 16), 16, 16)
    rng = np.random.default_rng(10)
    wi = rng.uniform(0.5, 1.5, (ro.shape[0], 3)).astype(np.float32)
    out["trn_rays_o"], out["trn_rays_d"], out["trn_w_image"] = ro, rd, wi
    for tag, kw in (("trn", dict(perturb=False)), ("trnp", dict(perturb=True))):
        for p in field.parameters():
            p.grad = None
        res = ren.run_cuda(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], dt_gamma=0, bg_color=1.0, force_all_rays=False,
                           max_steps=1024, **kw)
        (res["image"][0] * torch.from_numpy(wi)).sum().backward()
        out[f"{tag}_image"], out[f"{tag}_depth"], out[f"{tag}_weights_sum"] = _np(res["image"][0]), _np(res["depth"][0]), _np(res["weights_sum"])
        out[f"{tag}_counter"] = ren.step_counter.numpy().copy()
        out[f"{tag}_local_step"] = np.int64(ren.local_step)
        for k, w in enumerate(field.sigma_weights):
            out[f"{tag}_grad_sigma_w{k}"] = _np(w.grad)
        for k, w in enumerate(field.color_weights):
            out[f"{tag}_grad_color_w{k}"] = _np(w.grad)
        for k, v in table_grad_digest(_np(field.embeddings.grad)).items():
            out[f"{tag}_grad_table_{k}"] = v
    # the mean_count feedback of update_extra_state's tail (nerf/renderer.py:534-537) and the bounded allocation it causes
    # (raymarching/raymarching.py:196-203): third call with mean_count known
    total = min(16, ren.local_step)
    ren.mean_count = int(ren.step_counter[:total, 0].sum().item() / total)
    out["trn3_mean_count"] = np.int64(ren.mean_count)
    res = ren.run_cuda(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], dt_gamma=0, bg_color=1.0, perturb=False, max_steps=1024)
    out["trn3_image"], out["trn3_counter"] = _np(res["image"][0].detach()), ren.step_counter.numpy().copy()
    return out


def tier2_grid(R, H=32, seed=0):
    """update_extra_state (full sweep, then a partial one) and mark_untrained_grid on an H^3 grid"""
    # one thread: `tmp_grid[cas, indices] = sigmas` (nerf/renderer.py:486,516) writes duplicate indices in an order that depends on the thread count;
    # with one thread the LAST writer wins and the fixture is reproducible (the tests accept any writer on those cells)
    torch.set_num_threads(1)
    model, field = oracle_field()
    ren = small_grid(make_renderer(R, field, bound=W.BOUND, cuda_ray=True, min_near=0.2, density_thresh=10), H)
    cas = ren.cascade
    out = {"H": np.int64(H), "seed": np.int64(seed)}

    # mark_untrained_grid: five cameras close to the scene, so that part of the grid is unseen
    poses = np.stack([W.orbit_pose(k, n=5, radius=1.2, height=0.4) for k in range(5)])
    intr = W.intrinsics(64, 64)
    ren.mark_untrained_grid(poses, intr, S=16)
    out["mark_poses"], out["mark_intrinsics"] = poses, intr
    out["mark_unseen"] = np.packbits(ren.density_grid.numpy() < 0)
    ren.density_grid.zero_()

    def replacements(rnd, partial):
        calls = {"rand_like": 0, "randint": 0}
        coords_mesh = None
        if not partial:
            ar = torch.arange(H, dtype=torch.int32)
            xx, yy, zz = torch.meshgrid(ar, ar, ar, indexing="ij")
            coords_mesh = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
            order = O.morton3D(coords_mesh.numpy()).astype(np.int64)

        def rand_like(t, **k):
            c = calls["rand_like"]; calls["rand_like"] += 1
            if not partial:                                           # one call per cascade, rows in custom_meshgrid order (:460-481)
                return _t(rnd["noise"][c][order]).to(t.dtype)
            n_occ = t.shape[0] - rnd["noise_rand"][c].shape[0]
            both = np.concatenate([rnd["noise_rand"][c], rnd["noise_occ"][c][:n_occ]])
            return _t(both).to(t.dtype)

        def randint(lo, hi, size, **k):
            c = calls["randint"]; calls["randint"] += 1
            cascade_i, which = divmod(c, 2)
            if which == 0:                                            # coords (:494)
                return _t(rnd["coords"][cascade_i].astype(np.int64))
            return _t(rnd["pick"][cascade_i].astype(np.int64))        # rand_mask (:498), index into the occupied list
        return {"rand_like": rand_like, "randint": randint}

    # full sweep from an all-zero grid; two training steps are on the counter ring
    ren.step_counter[0, 0], ren.step_counter[1, 0], ren.local_step = 1000, 1301, 2
    rnd = CO.grid_update_randoms(seed, ren.iter_density, cas, H, partial=False)
    with Capture(**replacements(rnd, False)):
        ren.update_extra_state(decay=0.95, S=H)
    out["full_grid"], out["full_bitfield"] = ren.density_grid.numpy().copy(), ren.density_bitfield.numpy().copy()
    out["full_mean_density"], out["full_mean_count"] = np.float64(ren.mean_density), np.int64(ren.mean_count)

    # partial sweep (iter_density >= 16)
    ren.iter_density = 16
    grid_before = ren.density_grid.numpy().copy()
    n_occ = [(grid_before[c] > 0).sum() for c in range(cas)]
    rnd = CO.grid_update_randoms(seed, ren.iter_density, cas, H, partial=True, n_occ=n_occ)
    with Capture(**replacements(rnd, True)):
        ren.update_extra_state(decay=0.95, S=H)
    out["partial_grid"], out["partial_bitfield"] = ren.density_grid.numpy().copy(), ren.density_bitfield.numpy().copy()
    out["partial_mean_density"] = np.float64(ren.mean_density)
    torch.set_num_threads(8)
    return out


def tier3_fields():
    """NeRFNetwork of nerf/network.py (M1: bias-free Linear layers, optional background model) and of nerf/network_ff.py (M2: FFMLP wiring)"""
    import nerf.network as NW
    import nerf.network_ff as NF
    assert NW.__file__.startswith(REF) and NF.__file__.startswith(REF)
    model = W.make_model(0)
    rng = np.random.default_rng(21)
    x = rng.uniform(-2, 2, (3000, 3)).astype(np.float32)
    x[:5] = [[2, 2, 2], [-2, -2, -2], [0, 0, 0], [2, -2, 0.5], [0, 0, -2]]
    d = rng.normal(size=(3000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    mask = rng.uniform(size=3000) < 0.3
    out = {"x": x, "d": d, "mask": mask}
    # ---- M1: the default network, with its background model
    net = NW.NeRFNetwork(bound=W.BOUND, cuda_ray=False, bg_radius=3.0).eval()
    shapes = [tuple(l.weight.shape) for l in list(net.sigma_net) + list(net.color_net) + list(net.bg_net)]
    assert shapes == [(64, 32), (16, 64), (64, 31), (64, 64), (3, 64), (64, 24), (3, 64)], shapes
    with torch.no_grad():
        net.encoder.embeddings.copy_(torch.from_numpy(model["embeddings"]))
        net.encoder_bg.embeddings.copy_(torch.from_numpy(np.random.default_rng(22).uniform(-1, 1, tuple(net.encoder_bg.embeddings.shape)).astype(np.float32)))
        for k, l in enumerate(list(net.sigma_net) + list(net.color_net) + list(net.bg_net)):
            w = rng.uniform(-0.4, 0.4, tuple(l.weight.shape)).astype(np.float32)
            l.weight.copy_(torch.from_numpy(w))
            out[f"m1_w{k}"] = w
    out["m1_bg_table_seed"], out["m1_bg_table_shape"] = np.int64(22), np.array(net.encoder_bg.embeddings.shape)
    out["m1_offsets"], out["m1_bg_offsets"] = net.encoder.offsets.numpy(), net.encoder_bg.offsets.numpy()
    out["m1_per_level_scale"], out["m1_bg_per_level_scale"] = np.float64(net.encoder.per_level_scale), np.float64(net.encoder_bg.per_level_scale)
    tx, td = torch.from_numpy(x), torch.from_numpy(d)
    with torch.no_grad():
        sigma, color = net(tx, td)
        dens = net.density(tx)
        cm = net.color(tx, td, mask=torch.from_numpy(mask), **dens)
        sph = torch.from_numpy(O.sph_from_ray(x * 0.1, d, 3.0))
        bg = net.background(sph, td)
    out.update(m1_sigma=_np(sigma), m1_color=_np(color), m1_density_sigma=_np(dens["sigma"]), m1_geo_feat=_np(dens["geo_feat"]), m1_color_masked=_np(cm),
               m1_sph=_np(sph), m1_background=_np(bg), m1_n_param_groups=np.int64(len(net.get_params(1e-2))))
    # gradient of a weighted sum w.r.t. the points (the nav loop differentiates density w.r.t. x: simulate.py:343, nav/quad_plot.py:237)
    xg = tx.clone().requires_grad_(True)
    out["m1_w_sum"] = rng.uniform(0.5, 1.5, 3000).astype(np.float32)
    (net.density(xg)["sigma"] * torch.from_numpy(out["m1_w_sum"])).sum().backward()
    out["m1_grad_x"] = _np(xg.grad)
    # ---- M2: the FFMLP network with the hand-set S-ring model
    nf = NF.NeRFNetwork(bound=W.BOUND, cuda_ray=False).eval()
    assert nf.in_dim_color == 32 and nf.sigma_net.weights.numel() == 7168 and nf.color_net.weights.numel() == 11264
    with torch.no_grad():
        nf.encoder.embeddings.copy_(torch.from_numpy(model["embeddings"]))
        nf.sigma_net.weights.copy_(torch.from_numpy(model["sigma_weights"]))
        nf.color_net.weights.copy_(torch.from_numpy(model["color_weights"]))
        sigma, rgb = nf(tx, td)
        dens = nf.density(tx)
        cm = nf.color(tx, td, mask=torch.from_numpy(mask), **dens)
    out.update(m2_sigma=_np(sigma), m2_rgb=_np(rgb), m2_geo_feat=_np(dens["geo_feat"]), m2_color_masked=_np(cm))
    return out


def nav_network():
    """the reference's default NeRFNetwork (nerf/network.py) holding the S-ring nav model (ngp.workload.make_model(0) table + nav_weights(0)), eval mode"""
    import nerf.network as NW
    assert NW.__file__.startswith(REF)
    model = W.make_model(0)
    sw, cw = W.nav_weights(0)
    net = NW.NeRFNetwork(bound=W.BOUND, cuda_ray=False).eval()
    with torch.no_grad():
        net.encoder.embeddings.copy_(torch.from_numpy(model["embeddings"]))
        for layer, w in zip(list(net.sigma_net) + list(net.color_net), sw + cw):
            assert tuple(layer.weight.shape) == w.shape
            layer.weight.copy_(torch.from_numpy(w))
    return net


NAV_ROT = [[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]            # simulate.py:340


def nav_planner_cfg(steps=8, nbins=(4, 4, 3)):
    """simulate.py:266-283 with a shorter horizon and a coarser body (the fixture stays small)"""
    return {"T_final": 2., "steps": steps, "lr": 0.001, "epochs_init": 1, "fade_out_epoch": 0, "fade_out_sharpness": 10, "epochs_update": 1,
            "I": torch.eye(3), "g": 10., "mass": 1., "body": np.array([[-0.05, 0.05], [-0.05, 0.05], [-0.02, 0.02]]), "nbins": list(nbins)}


def tier4_nav(U):
    """The two nav/ callers of the hot path, EXECUTED: Planner (nav/quad_plot.py:10-60,120-254) and Estimator.measurement_fn + the Hessian call
    (nav/estimator_helpers.py:127-170,293-327,384) over simulate.py:340-347's three lambdas on the reference's default NeRFNetwork.
    cv2 / imageio are empty placeholder modules (nav/estimator_helpers.py:5, nav/agent_helpers.py:5,7 import them; nothing on these paths uses them)."""
    import nav.estimator_helpers as EH
    import nav.math_utils as MU
    import nav.quad_plot as QP
    assert QP.__file__.startswith(REF) and EH.__file__.startswith(REF)
    ENCODER_RULE["first_order"] = True
    net = nav_network()
    rot = torch.tensor(NAV_ROT)
    density_fn = lambda x: net.density(x.reshape((-1, 3)) @ rot)["sigma"].reshape(x.shape[:-1])                      # noqa: E731  simulate.py:343
    out = {}

    # ---- (a) planner: cost of a near-straight trajectory THROUGH the pillar at 180 degrees (NeRF frame (-0.65, 0, 0..0.45) = planner frame (z, x, y)), gradient w.r.t. its parameters
    start = torch.cat([torch.tensor([0.2, -0.9, -0.05]), torch.zeros(3), MU.vec_to_rot_matrix(torch.tensor([0., 0., 0.])).reshape(-1), torch.zeros(3)])
    end = torch.cat([torch.tensor([0.25, -0.3, 0.05]), torch.zeros(3), MU.vec_to_rot_matrix(torch.tensor([0., 0., 0.])).reshape(-1), torch.zeros(3)])
    for tag, cfg, epoch in (("pl", nav_planner_cfg(), 0), ("plf", dict(nav_planner_cfg(steps=10, nbins=(3, 3, 2)), fade_out_epoch=8), 3)):
        pl = QP.Planner(start, end, cfg, density_fn)
        pl.epoch = epoch
        with torch.no_grad():                                        # leave the straight line a little so that no term is degenerate
            pl.states += torch.from_numpy(np.random.default_rng(41).normal(scale=0.02, size=tuple(pl.states.shape)).astype(np.float32))
            pl.initial_accel += torch.tensor([0.3, -0.2])
        per_state, collision = pl.get_state_cost()
        total = pl.total_cost()
        total.backward()
        pts = pl.body_to_world(pl.robot_body).detach()
        out.update({f"{tag}_start": _np(start), f"{tag}_end": _np(end), f"{tag}_steps": np.int64(cfg["steps"]), f"{tag}_nbins": np.array(cfg["nbins"]),
                    f"{tag}_epoch": np.int64(epoch), f"{tag}_fade_out_epoch": np.int64(cfg["fade_out_epoch"]),
                    f"{tag}_states": _np(pl.states), f"{tag}_initial_accel": _np(pl.initial_accel), f"{tag}_robot_body": _np(pl.robot_body),
                    f"{tag}_points": _np(pts), f"{tag}_sigma": _np(density_fn(pts)), f"{tag}_per_state": _np(per_state), f"{tag}_collision": _np(collision),
                    f"{tag}_total": _np(total), f"{tag}_grad_states": _np(pl.states.grad), f"{tag}_grad_initial_accel": _np(pl.initial_accel.grad),
                    f"{tag}_actions": _np(pl.get_actions()), f"{tag}_full_states": _np(pl.get_full_states())})
    assert float(out["pl_collision"].max()) > 1.0, "the trajectory does not touch the scene"

    # ---- (b) pose filter: measurement_fn's loss, gradient and the 12 x 12 Hessian exactly as estimate_state asks for it (:384)
    Hh, Wd, steps = 20, 20, 64
    intr = W.intrinsics(Hh, Wd)
    get_rays_fn = lambda pose: U.get_rays(pose, intr, Hh, Wd)                                                            # noqa: E731  simulate.py:347
    render_fn = lambda rays_o, rays_d: net.render(rays_o, rays_d, staged=True, bg_color=1., perturb=False, num_steps=steps, upsample_steps=0)  # noqa: E731  :346
    rng = np.random.default_rng(42)
    A = rng.normal(size=(12, 12)).astype(np.float32)
    sig = torch.from_numpy(np.eye(12, dtype=np.float32) + 0.05 * A @ A.T)
    cfg = {"dil_iter": 3, "batch_size": 16, "kernel_size": 5, "lrate": 1e-3, "N_iter": 1, "sig0": sig, "Q": torch.eye(12), "render_viz": False, "show_rate": [20, 100]}
    x_prev = torch.tensor([0.95, -1.1, 0.45, 0.1, -0.05, 0.02, 0.12, -0.2, 2.3, 0.01, 0.02, -0.03])
    est = EH.Estimator(cfg, None, x_prev.clone(), filter=True, get_rays_fn=get_rays_fn, render_fn=render_fn)
    x = x_prev + torch.from_numpy(rng.normal(scale=0.02, size=12).astype(np.float32))
    target = torch.from_numpy(rng.uniform(0, 1, (Hh, Wd, 3)).astype(np.float32))
    batch = np.stack([rng.integers(0, Hh, 16), rng.integers(0, Wd, 16)], axis=-1)
    xs = x.clone().requires_grad_(True)
    loss = est.measurement_fn(xs, x_prev, sig, target, batch)
    loss.backward()
    hess = torch.autograd.functional.hessian(lambda s: est.measurement_fn(s, x_prev.clone().detach(), sig, target, batch), x.clone().detach())
    with torch.no_grad():
        rgb_loss = loss - MU.mahalanobis(x, x_prev, sig)
        view = est.render_from_pose(torch.cat([torch.cat([MU.vec_to_rot_matrix(x[6:9]), x[:3, None]], dim=1), torch.tensor([[0., 0., 0., 1.]])]))
    inv = torch.inverse(sig)
    out.update(mf_state=_np(x), mf_start=_np(x_prev), mf_sig=_np(sig), mf_target=_np(target), mf_batch=batch.astype(np.int64), mf_HW=np.array([Hh, Wd]),
               mf_intrinsics=intr, mf_num_steps=np.int64(steps), mf_loss=_np(loss), mf_rgb_loss=_np(rgb_loss), mf_grad=_np(xs.grad), mf_hessian=_np(hess),
               mf_hessian_process=_np(inv + inv.T), mf_view=_np(view))
    rgb_part = out["mf_hessian"] - out["mf_hessian_process"]
    assert np.abs(rgb_part[6:9, 6:9]).max() > 1e-4, "the rays do not see the scene: the image term of the Hessian vanishes"
    ENCODER_RULE["first_order"] = False
    return out


def tier5_composite(R):
    """SURVEY 8c relation 1 with the reference's side EXECUTED: what NeRFRenderer.run (nerf/renderer.py:206-230) feeds its own torch compositing --
    the exponent -delta * density_scale * sigma (:208, read off `torch.exp`), the colours (:218) -- and what it returns, so that the native
    compositor (composite_rays_train_forward) can be checked on the same numbers."""
    model, field = oracle_field()
    ren = make_renderer(R, field, bound=W.BOUND, cuda_ray=False, min_near=0.2, density_thresh=10).eval()
    ro, rd = run_rays()
    seen = {}
    orig_exp, orig_color = torch.exp, ren.color

    def exp_spy(x, *a, **k):
        if x.dim() == 2 and "exponent" not in seen:
            seen["exponent"] = x.detach().clone()
        return orig_exp(x, *a, **k)

    def color_spy(x, d, mask=None, **kw):
        rgbs = orig_color(x, d, mask=mask, **kw)
        seen["rgbs"], seen["mask"] = rgbs.detach().clone(), mask.detach().clone()
        return rgbs

    torch.exp, ren.color = exp_spy, color_spy
    try:
        with torch.no_grad():
            res = ren.run(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], bg_color=1.0, num_steps=64, upsample_steps=0)
    finally:
        torch.exp = orig_exp
    N = ro.shape[0]
    assert seen["exponent"].shape == (N, 64) and seen["rgbs"].shape == (N * 64, 3)
    return {"rays_o": ro, "rays_d": rd, "exponent": _np(seen["exponent"]), "rgbs": _np(seen["rgbs"]).reshape(N, 64, 3), "mask": _np(seen["mask"]).reshape(N, 64),
            "image": _np(res["image"][0]), "weights_sum": _np(res["weights_sum"]), "depth": _np(res["depth"][0])}


ONLY = set(sys.argv[1:])                       # e.g. `... make_callers_golden.py callers_nav` regenerates that file alone


def main():
    U, R, P = import_reference()
    torch.set_num_threads(8)
    O.set_threads(8)
    for name, fn in (("callers_tier1", lambda: tier1(U, R, P)), ("callers_run", lambda: tier2_run(R)),
                     ("callers_run_cuda", lambda: tier2_run_cuda(R)), ("callers_grid", lambda: tier2_grid(R)), ("callers_fields", tier3_fields),
                     ("callers_nav", lambda: tier4_nav(U)), ("callers_composite", lambda: tier5_composite(R))):
        if ONLY and name not in ONLY:
            continue
        data = fn()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{name}: {len(data)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")
    for sub in ("nerf", "nav"):
        assert not os.path.exists(os.path.join(REF, sub, "__pycache__")), "bytecode was written into the reference tree"


if __name__ == "__main__":
    main()
