"""What the HOST needs per training step: the steady-state step with 64 rays instead of 4,096 (the GPU work shrinks to the launch floor, the Python / runtime work
per step stays), direct step and autograd step.   python tools/time_train_host.py"""
import importlib, os, sys, time, gc
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
from ngp import workload as W
from ngp.field import NGPFieldFF
from ngp.render import NGPRenderer
from ngp.train import NGPTrainer
dev = torch.device("cuda:0")
teacher = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0)), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
teacher.load_density_grid(W.density_grid())
res = 200
o, d = W.get_rays(W.orbit_pose(0, 8), W.intrinsics(res, res), res, res)
to, td = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
tc = teacher.render_fused(to, td, bg_color=1, image_width=res)["image"]
for direct in (True, False):
    torch.manual_seed(0)
    student = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev)
    tr = NGPTrainer(student, lr=1e-2, iters=30000, fp16=True, direct=direct)
    gen = torch.Generator(device=dev).manual_seed(1)
    def step(n_rays):
        idx = torch.randint(0, res * res, (n_rays,), device=dev, generator=gen)
        return tr.step(to[:, idx], td[:, idx], tc[:, idx], bg_color=1, max_steps=1024)
    for k in range(1500):
        step(4096)
    for n_rays in (4096, 64):
        for k in range(48):
            step(n_rays)                              # mean_count follows the batch size
        torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter()
        for k in range(200):
            if (tr.global_step % 16) == 0:
                pass
            step(n_rays)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        gc.enable()
        print(f"direct={direct} rays={n_rays}: loss {float(step(n_rays)):.5f} host {1e3 * (t1 - t0) / 200:.3f} ms/step, wall {1e3 * (t2 - t0) / 200:.3f} ms/step")
