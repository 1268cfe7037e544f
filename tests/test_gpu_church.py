"""GPU: BASELINE config 5's scene ("church": ngp/workload.py church_boxes, the larger shell scene of SURVEY 8d row 5) on the HIP path against
the CPU oracle, at sizes the oracle finishes in seconds: the fused frame, the per-op run_cuda loop, and one 4,096-ray training step (what
every rank of the 8-GPU job runs before its gradient all-reduce).  The multi-GPU part of config 5 is covered by the 2-rank gloo tests
(tests/test_distributed_cpu.py) and the single-rank RCCL test (tests/test_gpu_rccl_single_rank.py)."""
import numpy as np
import pytest
import torch

from oracle import render_oracle as R

pytestmark = pytest.mark.gpu
HW = 48


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def church(oracle, dev):
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    model = W.make_model(0, scene="church")
    grid = W.density_grid(scene="church")
    bitfield, _ = W.bitfield_from_grid(grid)
    field = NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(model)
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
    ren.load_density_grid(grid)
    assert np.array_equal(ren.density_bitfield.cpu().numpy(), bitfield)
    radius, height = W.scene_orbit("church")
    o, d = W.get_rays(W.orbit_pose(1, 8, radius, height), W.intrinsics(HW, HW), HW, HW)
    return dict(W=W, model=model, bitfield=bitfield, field=field, ren=ren, o=o, d=d, grid=grid)


def test_church_fused_frame_vs_oracle(church, dev):
    """config 5 render, one launch: per-ray sample counts, image, depth against render_single_march (same bars as the ring scene)"""
    ren, model = church["ren"], church["model"]
    ref = R.render_single_march(lambda x, d: R.field_forward(model, x, d, 1.0), church["o"], church["d"], church["bitfield"], 2.0, 2)
    out = ren.render_fused(t(church["o"], dev)[None], t(church["d"], dev)[None], dt_gamma=0, bg_color=1, max_steps=1024, image_width=HW)
    stats = out["stats"].cpu().numpy()
    img = out["image"][0].cpu().numpy()
    assert ref["samples"] > 1.3 * 44 * HW * HW * 0.5                     # the shell scene is the heavier one (66 vs 45 samples per ray at 800x800)
    assert abs(int(stats[0]) - ref["samples"]) <= max(8, 2e-4 * ref["samples"]), (stats, ref["samples"])
    assert stats[1] == 0 and stats[2] == int((ref["consumed"] > 0).sum())
    assert np.max(np.abs(img - ref["image"])) < 5e-3 and R.psnr(img, ref["image"]) > 60
    dep = out["depth"][0].cpu().numpy()
    ok = np.isfinite(ref["depth"])
    assert np.array_equal(np.isfinite(dep), ok) and np.max(np.abs(dep[ok] - ref["depth"][ok])) < 2e-3


def test_church_per_op_loop_vs_oracle(church, dev):
    """config 5 render through the drop-in ops, one by one: the alive-count schedule and the image against the oracle's run_cuda"""
    ren, model = church["ren"], church["model"]
    tr_ref, tr = [], []
    ref = R.run_cuda(lambda x, d: R.field_forward(model, x, d, 1.0), church["o"], church["d"], church["bitfield"], 2.0, 2, trace=tr_ref)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out = ren.run_cuda(t(church["o"], dev)[None], t(church["d"], dev)[None], dt_gamma=0, bg_color=1, perturb=False, max_steps=1024, trace=tr,
                           fused_field=False)
    same = sum(a == b for a, b in zip(tr, tr_ref))
    assert len(tr) == len(tr_ref) and same >= len(tr_ref) - 3, (len(tr), len(tr_ref), same)
    img = out["image"][0].cpu().numpy()
    assert np.max(np.abs(img - ref["image"])) < 5e-3 and R.psnr(img, ref["image"]) > 60


def test_church_training_step_vs_oracle(church, dev):
    """config 5, one rank's share of a step: 4,096 rays of a church view, perturb on, FFMLP field under autocast through the native training
    launches, against callers_oracle.run_cuda_train in float32 with the same master weights: counter bit-exact, image 4e-3, weight gradients
    3e-2 in norm, table gradient 5e-2 in norm (half activations and half atomics)."""
    from _util import ff_grads_as_matrices, oracle_field
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    from oracle import callers_oracle as CO
    W = church["W"]
    torch.manual_seed(6)
    field = NGPFieldFF(bound=W.BOUND).to(dev)
    with torch.no_grad():
        field.encoder.embeddings.uniform_(-0.3, 0.3)
        field.sigma_net.weights.mul_(0.6)
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).train()
    ren.load_density_grid(church["grid"])
    radius, height = W.scene_orbit("church")
    o, d = W.get_rays(W.orbit_pose(3, 8, radius, height), W.intrinsics(64, 64), 64, 64)
    target = np.random.default_rng(3).uniform(0, 1, size=(4096, 3)).astype(np.float32)
    scale = 1024.0
    with torch.autocast("cuda", dtype=torch.float16):
        out = ren.run_cuda(t(o, dev)[None], t(d, dev)[None], dt_gamma=0, bg_color=1, perturb=True, force_all_rays=False, max_steps=1024)
        loss = torch.nn.functional.mse_loss(out["image"][0], t(target, dev))
    (loss * scale).backward()
    orc = oracle_field(field)
    ref = CO.run_cuda_train(orc, o, d, church["bitfield"], W.BOUND, 2, perturb=True, mean_count=0)
    torch.nn.functional.mse_loss(ref["image"], torch.from_numpy(target)).backward()
    cnt = ren.step_counter[0].cpu().numpy()
    assert np.array_equal(cnt, ref["counter"]) and cnt[1] == 4096 and cnt[0] > 100000
    assert np.max(np.abs(out["image"][0].detach().float().cpu().numpy() - ref["image"].detach().numpy())) < 4e-3
    gs = [g / scale for g in ff_grads_as_matrices(field.sigma_net)] + [g / scale for g in ff_grads_as_matrices(field.color_net)]
    for g, w in zip(gs, orc.sigma_weights + orc.color_weights):
        assert rel(g, w.grad.numpy()) < 3e-2
    assert rel(field.encoder.embeddings.grad.float().cpu().numpy() / scale, orc.embeddings.grad.numpy()) < 5e-2
