"""How much of the table-gradient scatter's input is exactly zero?  (An entry whose two half values are zero adds nothing to the exact sums.)
Runs the trainer of `bench.py --mode train` into its steady state and reports, for a few steps: the share of samples whose incoming gradients are
all zero (behind the compositor's early exit, or in slots the march did not fill), and the share of (level, sample) pairs with a zero feature gradient.
    python tools/zero_grad_fraction.py [--settle 1500]"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--settle", type=int, default=1500)
    args = ap.parse_args()
    from gridencoder import grid as G
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    from ngp.train import NGPTrainer
    dev = torch.device("cuda:0")
    scene = "ring"
    field = NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0, scene=scene))
    teacher = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
    teacher.load_density_grid(W.density_grid(scene=scene))
    res, n_rays = 200, 4096
    intr = W.intrinsics(res, res)
    radius, height = W.scene_orbit(scene)
    pool = []
    for view in range(8):
        o, d = W.get_rays(W.orbit_pose(view, 8, radius, height), intr, res, res)
        to, td = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
        pool.append((to, td, teacher.render_fused(to, td, bg_color=1, image_width=res)["image"]))
    torch.manual_seed(0)
    student = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev)
    tr = NGPTrainer(student, lr=1e-2, iters=30000, fp16=True, steps_per_epoch=len(pool))
    gen = torch.Generator(device=dev).manual_seed(1)
    stats = []
    inner = G.table_gradient_binned

    def spy(grad_enc, inputs, offsets, M, L, *a, **k):
        if spy.on:
            g = grad_enc.view(L, M, -1)
            zero_pair = (g == 0).all(-1)                                  # [L, M]
            stats.append((M, float(zero_pair.float().mean()), float(zero_pair.all(0).float().mean()), [float(v) for v in zero_pair.float().mean(1)]))
        return inner(grad_enc, inputs, offsets, M, L, *a, **k)
    spy.on = False
    G.table_gradient_binned = spy

    def step(k):
        to, td, tc = pool[k % len(pool)]
        idx = torch.randint(0, res * res, (n_rays,), device=dev, generator=gen)
        return tr.step(to[:, idx], td[:, idx], tc[:, idx], bg_color=1, max_steps=1024)
    for phase, first, n in (("early", 20, 4), ("steady", args.settle, 4)):
        for k in range(len(stats) and 24, first):
            step(k)
        spy.on = True
        for k in range(n):
            step(first + k)
        spy.on = False
        torch.cuda.synchronize()
        for M, pairs, samples, per_level in stats[-n:]:
            print(f"{phase}: M {M}  used {int(student.step_counter[(student.local_step - 1) % 16, 0])}  zero (level, sample) pairs {pairs:.3f}  samples with all 16 levels zero {samples:.3f}  "
                  f"per level {' '.join('%.2f' % v for v in per_level)}")


if __name__ == "__main__":
    main()
