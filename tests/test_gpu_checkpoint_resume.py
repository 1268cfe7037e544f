"""GPU: a full checkpoint in the reference's layout (nerf/utils.py:938-986: epoch, global_step, stats, mean_count, mean_density, optimizer, lr_scheduler,
scaler, ema, model) written from a running NGPTrainer -- native optimiser and all -- read back with weights_only=True into a fresh renderer + trainer,
which then continues like the original; and the same file's optimiser / scaler entries loaded into the torch classes the reference uses."""
import importlib

import numpy as np
import pytest
import torch

importlib.import_module("nerf-navigation_amd")
pytestmark = pytest.mark.gpu


def _setup(dev, seed):
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    from ngp.train import NGPTrainer
    torch.manual_seed(seed)
    student = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev)
    return student, NGPTrainer(student, lr=1e-2, iters=500, fp16=True, steps_per_epoch=8)


def test_full_checkpoint_round_trip_and_resume(dev, tmp_path):
    from ngp import checkpoint as CK
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    teacher = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0)), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
    teacher.load_density_grid(W.density_grid())
    res, n_rays = 48, 1024
    o, d = W.get_rays(W.orbit_pose(2, 8), W.intrinsics(res, res), res, res)
    to, td = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
    tc = teacher.render_fused(to, td, bg_color=1, image_width=res)["image"]

    def batch(gen):
        idx = torch.randint(0, res * res, (n_rays,), device=dev, generator=gen)
        return to[:, idx], td[:, idx], tc[:, idx]

    a, tra = _setup(dev, 0)
    gen = torch.Generator(device=dev).manual_seed(1)
    for _ in range(20):
        tra.step(*batch(gen), bg_color=1, max_steps=256)
    path = str(tmp_path / "ngp_ep0003.pth")
    CK.write_checkpoint(path, a, tra, epoch=3, stats={"loss": [0.1]})
    blob = torch.load(path, map_location="cpu", weights_only=True)                     # nothing in the file needs executing
    assert set(blob) == {"epoch", "global_step", "stats", "mean_count", "mean_density", "optimizer", "lr_scheduler", "scaler", "ema", "model"}
    assert blob["global_step"] == 20 and blob["epoch"] == 3 and blob["scaler"]["scale"] == 65536.0 and blob["ema"]["num_updates"] == 2
    assert {"encoder.embeddings", "sigma_net.weights", "color_net.weights", "density_grid", "density_bitfield", "step_counter"} <= set(blob["model"])
    # the torch classes the reference uses take the optimiser and scaler entries as they are
    ref_opt = torch.optim.Adam(a.field.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    ref_opt.load_state_dict(blob["optimizer"])
    assert int(ref_opt.state[a.field.encoder.embeddings]["step"]) == 20
    ref_scaler = torch.amp.GradScaler("cuda")
    ref_scaler.load_state_dict(blob["scaler"])
    assert ref_scaler.get_scale() == 65536.0
    # a fresh renderer + trainer (different initial weights) resumes and continues like the original
    b, trb = _setup(dev, 123)
    assert CK.resume_trainer(path, b, trb) == 3
    assert trb.global_step == 20 and trb.opt.step_count() == 20 and trb.ema.num_updates == 2 and b.mean_count == a.mean_count
    for p, q in zip(a.field.parameters(), b.field.parameters()):
        assert torch.equal(p, q)
    assert torch.equal(a.density_bitfield, b.density_bitfield) and trb.sched.get_last_lr() == tra.sched.get_last_lr()
    b.local_step, b.iter_density = a.local_step, a.iter_density                        # run-time counters the reference does not store either
    ga, gb = torch.Generator(device=dev).manual_seed(7), torch.Generator(device=dev).manual_seed(7)
    la = [float(tra.step(*batch(ga), bg_color=1, max_steps=256)) for _ in range(6)]
    lb = [float(trb.step(*batch(gb), bg_color=1, max_steps=256)) for _ in range(6)]
    assert np.allclose(la, lb, rtol=2e-3, atol=1e-7), (la, lb)
