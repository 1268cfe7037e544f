"""GPU: the callers either side of the hot path, at the reference's own sizes.
  config 3 (main_nerf.py training, nerf/utils.py:404-487,774-794): run_cuda's training branch through the drop-in ops with
           autograd -- FFMLP field under autocast + GradScaler, and the nn.Linear field in fp32;
  density-grid maintenance (nerf/renderer.py:446-537);
  config 4 (simulate.py nav loop): density_fn with gradient to the points (nav/quad_plot.py:237), run() with gradient to
           the rays (nav/estimator_helpers.py:316), and the reference's second-order semantics."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def scene(dev):
    from ngp import workload as W
    model = W.make_model(0)
    grid = W.density_grid()
    o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(64, 64), 64, 64)
    return dict(W=W, model=model, grid=grid, o=o, d=d)


def make(scene, dev, ff):
    from ngp.field import NGPField, NGPFieldFF
    from ngp.render import NGPRenderer
    W = scene["W"]
    field = (NGPFieldFF if ff else NGPField)(bound=W.BOUND).to(dev)
    if ff:
        field.load_arrays(scene["model"])
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev)
    ren.load_density_grid(scene["grid"])
    return ren


@pytest.mark.parametrize("ff", [True, False], ids=["ffmlp_autocast", "linear_fp32"])
def test_training_steps_reduce_the_loss(scene, dev, ff):
    ren = make(scene, dev, ff).train()
    o, d = t(scene["o"], dev)[None], t(scene["d"], dev)[None]            # 4096 rays, the reference's num_rays
    target = torch.full((1, 4096, 3), 0.25, device=dev)
    opt = torch.optim.Adam(ren.field.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)   # main_nerf.py:126
    scaler = torch.amp.GradScaler("cuda", enabled=ff)
    losses = []
    for it in range(12):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=ff):
            out = ren.run_cuda(o, d, dt_gamma=0, bg_color=1, perturb=True, force_all_rays=False, max_steps=1024)
            loss = torch.nn.functional.mse_loss(out["image"], target)
        scaler.scale(loss).backward()
        if it == 0:
            params = {n: p for n, p in ren.field.named_parameters()}
            for n, p in params.items():
                assert p.grad is not None and torch.isfinite(p.grad).all(), n
                assert p.grad.abs().sum() > 0, n
            cnt = ren.step_counter[0].cpu().numpy()
            assert cnt[1] == 4096 and cnt[0] > 10000                       # rays and points of the first step
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss))
    assert losses[-1] < 0.7 * losses[0], losses
    assert ren.local_step == 12


def test_mean_count_feedback_and_update_extra_state(scene, dev):
    ren = make(scene, dev, True).train()
    o, d = t(scene["o"], dev)[None], t(scene["d"], dev)[None]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for _ in range(3):
            ren.run_cuda(o, d, perturb=True)
        before = ren.density_bitfield.clone()
        ren.update_extra_state()                                            # full sweep: 2 x 128^3 density queries
    assert ren.iter_density == 1 and ren.local_step == 0
    assert ren.mean_count > 10000                                           # average points per step of the counter ring
    # the field represents the scene, so the re-estimated occupancy must agree with the analytic one almost everywhere
    a = np.unpackbits(before.cpu().numpy(), bitorder="little").astype(bool)
    b = np.unpackbits(ren.density_bitfield.cpu().numpy(), bitorder="little").astype(bool)
    assert (a & b).sum() > 0.6 * a.sum()
    # with mean_count known the next march allocates mean_count rounded up past 128 instead of N * max_steps
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out = ren.run_cuda(o, d, perturb=True)
    assert torch.isfinite(out["image"]).all()


def test_mark_untrained_grid(scene, dev):
    ren = make(scene, dev, True)
    W = scene["W"]
    poses = np.stack([W.orbit_pose(k) for k in range(8)])
    intr = W.intrinsics(64, 64)
    ren.density_grid.fill_(1.0)
    ren.mark_untrained_grid(poses, intr)
    g = ren.density_grid.cpu().numpy()
    frac = (g == -1).mean()
    assert 0.0 < frac < 0.9                                                 # some cells are seen by no camera, most of the centre is
    centre = W.morton3(np.array([64]), np.array([64]), np.array([64]))[0]
    assert g[0, centre] == 1.0


def test_nav_density_gradient_and_run_gradient(scene, dev):
    """fp32, no autocast, as simulate.py runs the field"""
    from ngp.field import NGPField
    from ngp.render import NGPRenderer
    W = scene["W"]
    torch.manual_seed(0)
    field = NGPField(bound=W.BOUND).to(dev)
    with torch.no_grad():
        field.encoder.embeddings.uniform_(-0.5, 0.5)
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=False).to(dev).eval()
    # density_fn (simulate.py:343): sigma and d sigma / d x on [S, 500, 3] body points
    rot = torch.tensor([[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]], device=dev)
    pts = (torch.rand(20, 500, 3, device=dev) * 2 - 1).requires_grad_(True)
    sigma = ren.density(pts.reshape(-1, 3) @ rot)["sigma"]
    sigma.sum().backward()
    assert pts.grad.shape == (20, 500, 3) and torch.isfinite(pts.grad).all() and pts.grad.abs().sum() > 0
    # finite-difference check of d sigma / d x on a few points: probe along the gradient direction, in float64 on the host side
    p0 = pts.detach().reshape(-1, 3)[:64].clone()
    g0 = pts.grad.reshape(-1, 3)[:64]
    direction = g0 / (g0.norm(dim=1, keepdim=True) + 1e-12)
    eps = 2e-5
    with torch.no_grad():
        sp = ren.density((p0 + eps * direction) @ rot)["sigma"].double()
        sm = ren.density((p0 - eps * direction) @ rot)["sigma"].double()
    fd = (sp - sm) / (2 * eps)
    an = (g0 * direction).sum(1).double()
    ok = an.abs() > 1e-2 * an.abs().max()                                   # away from cell faces the field is smooth
    rel = ((fd - an).abs() / an.abs().clamp_min(1e-6))[ok]
    assert rel.median() < 0.05, rel.median()

    # render_fn (simulate.py:346): run() with 1024 rays x 512 steps and gradient to rays_o / rays_d
    o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(32, 32), 32, 32)
    ro, rd = t(o, dev)[None].requires_grad_(True), t(d, dev)[None].requires_grad_(True)
    out = ren.render(ro, rd, staged=True, bg_color=1.0, perturb=False, num_steps=512, upsample_steps=0, max_ray_batch=4096)
    assert out["image"].shape == (1, 1024, 3)
    out["image"].sum().backward()
    assert torch.isfinite(ro.grad).all() and torch.isfinite(rd.grad).all() and rd.grad.abs().sum() > 0


def test_encoder_backward_is_constant_under_create_graph(dev):
    """N3: the reference's encoder backward calls a non-differentiable native op, so with create_graph=True the gradient
    w.r.t. the inputs carries no graph back to the inputs: Hessian terms through the encoders are silently zero
    (nav/estimator_helpers.py:384).  The drop-ins must keep that: no error, no double-backward implementation."""
    from gridencoder import GridEncoder
    from shencoder import SHEncoder
    enc = GridEncoder(num_levels=4, log2_hashmap_size=12, desired_resolution=64).to(dev)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x = (torch.rand(32, 3, device=dev) * 2 - 1).requires_grad_(True)
    y = enc(x, bound=1).pow(2).sum()
    (gx,) = torch.autograd.grad(y, x, create_graph=True)
    assert gx.abs().sum() > 0
    # the input gradient is produced by a native op inside backward: it carries NO graph, exactly as in the reference
    # (gridencoder/grid.py:81, @once_differentiable commented out at :62), so second derivatives through it vanish
    assert not gx.requires_grad and gx.grad_fn is None
    sh = SHEncoder(degree=4)
    v = torch.nn.functional.normalize(torch.randn(32, 3, device=dev), dim=1).requires_grad_(True)
    (gv,) = torch.autograd.grad(sh(v).pow(2).sum(), v, create_graph=True)
    assert gv.abs().sum() > 0 and not gv.requires_grad
    # what the pose filter calls (nav/estimator_helpers.py:384): functional.hessian fills the missing graph with zeros
    x0 = (torch.rand(4, 3, device=dev) * 2 - 1)
    H = torch.autograd.functional.hessian(lambda p: enc(p, bound=1).pow(2).sum(), x0)
    assert H.shape == (4, 3, 4, 3) and float(H.abs().max()) == 0.0


def test_trainer_fits_the_scene(scene, dev):
    """config 3 end to end: train a freshly initialised FFMLP field on views of the S-ring scene rendered by the
    hand-set model (the ground truth here), with the reference's recipe; PSNR on a held-out view must rise clearly."""
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    from ngp.train import NGPTrainer
    from oracle import render_oracle as R
    W = scene["W"]
    teacher = make(scene, dev, True).eval()
    res, intr = 64, W.intrinsics(64, 64)
    pool_o, pool_d, pool_c = [], [], []
    for k in range(12):
        o, d = W.get_rays(W.orbit_pose(k, 12), intr, res, res)
        to, td = t(o, dev)[None], t(d, dev)[None]
        pool_o.append(to); pool_d.append(td)
        pool_c.append(teacher.render_fused(to, td, bg_color=1)["image"])
    held = 5
    torch.manual_seed(0)
    student = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev)
    tr = NGPTrainer(student, lr=1e-2, iters=400, fp16=True)

    def psnr():
        with torch.no_grad():
            img = student.eval().render_fused(pool_o[held], pool_d[held], bg_color=1)["image"]
        return R.psnr(img.cpu().numpy(), pool_c[held].cpu().numpy())

    views = [k for k in range(12) if k != held]
    p0 = None
    for it in range(160):
        k = views[it % len(views)]
        loss = tr.step(pool_o[k], pool_d[k], pool_c[k], bg_color=1, max_steps=1024)
        if it == 0:
            p0 = psnr()
    p1 = psnr()
    assert torch.isfinite(loss) and student.mean_count > 0 and student.iter_density == 10
    assert p1 > p0 + 3.0 and p1 > 15.0, (p0, p1)
    # the trainer's fused Adam updates parameters without bumping `_version`: the packed copies must follow all the same
    student.field.fused_state(student.density_scale)
    emb_half = student.field._fused["tensors"][0]
    assert torch.equal(emb_half, student.field.encoder.embeddings.detach().half())
