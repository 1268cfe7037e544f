import importlib, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
import ngp_hip
from ngp import workload as W
from ngp.field import NGPFieldFF
from ngp.render import NGPRenderer
from ngp.train import NGPTrainer
dev = torch.device("cuda:0")
field = NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0, scene="ring"))
teacher = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
teacher.load_density_grid(W.density_grid(scene="ring"))
res, n_rays = 200, 4096
intr = W.intrinsics(res, res)
radius, height = W.scene_orbit("ring")
pool = []
for view in range(8):
    o, d = W.get_rays(W.orbit_pose(view, 8, radius, height), intr, res, res)
    to, td = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
    pool.append((to, td, teacher.render_fused(to, td, bg_color=1, image_width=res)["image"]))

def run(live, steps):
    ngp_hip.lib().ngp_field_train_set_live_only(live)
    torch.manual_seed(0)
    student = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev)
    tr = NGPTrainer(student, lr=1e-2, iters=30000, fp16=True, steps_per_epoch=len(pool))
    gen = torch.Generator(device=dev).manual_seed(1)
    losses = []
    for k in range(steps):
        to, td, tc = pool[k % len(pool)]
        idx = torch.randint(0, res * res, (n_rays,), device=dev, generator=gen)
        losses.append(tr.step(to[:, idx], td[:, idx], tc[:, idx], bg_color=1, max_steps=1024))
    return torch.stack([l.detach().float() for l in losses]).cpu()

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
a, b = run(0, steps), run(1, steps)
for k in list(range(0, 40, 4)) + list(range(40, steps, 50)):
    print(k, float(a[k]), float(b[k]), float(b[k] / a[k]))
for lo in range(0, steps, 200):
    print("mean loss steps", lo, lo + 200, float(a[lo:lo + 200].mean()), float(b[lo:lo + 200].mean()))
