"""A longer fit than bench.py's 2,000 steps: the loss scale's growth / backoff recurrence (GradScaler: doubles every 2,000 unskipped steps) running natively on the
device (ngp/optim.py), skipped steps, PSNR on a held-out view and samples per ray as the occupancy grid sharpens.   python tools/long_fit.py [--steps 8000]"""
import argparse, importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
from ngp import workload as W
from ngp.field import NGPFieldFF
from ngp.render import NGPRenderer
from ngp.train import NGPTrainer
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=8000)
ap.add_argument("--native", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda:0")
teacher = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0)), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
teacher.load_density_grid(W.density_grid())
res, n_rays = 200, 4096
pool = []
for view in range(24):
    o, d = W.get_rays(W.orbit_pose(view, 24, 1.6, 0.6 + 0.2 * ((view % 3) - 1)), W.intrinsics(res, res), res, res)
    to, td = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
    pool.append((to, td, teacher.render_fused(to, td, bg_color=1, image_width=res)["image"]))
o, d = W.get_rays(W.orbit_pose(5, 8), W.intrinsics(res, res), res, res)
ho, hd = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
held = teacher.render_fused(ho, hd, bg_color=1, image_width=res)["image"]
torch.manual_seed(0)
student = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev)
tr = NGPTrainer(student, lr=1e-2, iters=args.steps, fp16=True, steps_per_epoch=len(pool), native_adam=bool(args.native))
gen = torch.Generator(device=dev).manual_seed(1)
t0 = time.perf_counter()
for k in range(args.steps):
    to, td, tc = pool[k % len(pool)]
    idx = torch.randint(0, res * res, (n_rays,), device=dev, generator=gen)
    loss = tr.step(to[:, idx], td[:, idx], tc[:, idx], bg_color=1, max_steps=1024)
    if (k + 1) % 1000 == 0:
        torch.cuda.synchronize()
        student.eval()
        with torch.no_grad(), tr.eval_weights():
            out = student.render_fused(ho, hd, bg_color=1, image_width=res)
        student.train()
        mse = float(((out["image"] - held) ** 2).mean())
        steps_done = tr.opt.step_count() if tr.native_adam else int(next(iter(tr.opt.state.values()))["step"])
        print(f"step {k + 1}: loss {float(loss):.6f}  scale {tr.scaler.get_scale():.0f}  optimiser steps {steps_done} (skipped {k + 1 - steps_done})  "
              f"held-out PSNR {-10 * np.log10(mse):.2f} dB  samples/ray {float(out['stats'][0]) / (res * res):.1f}  {1e3 * (time.perf_counter() - t0) / (k + 1):.3f} ms/step", flush=True)
