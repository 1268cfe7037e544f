"""Where the HOST spends a training step (bench.py --mode train's step in its steady state): with the GPU side at ~0.85 ms per step the ~0.85 ms the host needs
to queue a step is the next bound.  cProfile over N steps after the grid has converged, synchronised only at the ends.
   python tools/profile_train_host.py [--steps 200] [--settle 600]"""
import argparse, cProfile, gc, importlib, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
from ngp import workload as W
from ngp.field import NGPFieldFF
from ngp.render import NGPRenderer
from ngp.train import NGPTrainer
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--settle", type=int, default=600)
ap.add_argument("--top", type=int, default=45)
args = ap.parse_args()
dev = torch.device("cuda:0")
teacher = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0)), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
teacher.load_density_grid(W.density_grid())
res, n_rays = 200, 4096
pool = []
radius, height = W.scene_orbit("ring")
for view in range(8):
    o, d = W.get_rays(W.orbit_pose(view, 8, radius, height), W.intrinsics(res, res), res, res)
    to, td = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
    pool.append((to, td, teacher.render_fused(to, td, bg_color=1, image_width=res)["image"]))
torch.manual_seed(0)
student = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev)
tr = NGPTrainer(student, lr=1e-2, iters=30000, fp16=True, steps_per_epoch=len(pool))
gen = torch.Generator(device=dev).manual_seed(1)


def step(k):
    to, td, tc = pool[k % len(pool)]
    idx = torch.randint(0, res * res, (n_rays,), device=dev, generator=gen)
    return tr.step(to[:, idx], td[:, idx], tc[:, idx], bg_color=1, max_steps=1024)


for k in range(args.settle):
    step(k)
torch.cuda.synchronize()
gc.collect(); gc.disable()
t0 = time.perf_counter()
for k in range(args.steps):
    step(args.settle + k)
queued = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"unprofiled: host queues a step in {1e3 * queued / args.steps:.3f} ms, step {1e3 * (time.perf_counter() - t0) / args.steps:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for k in range(args.steps):
    step(args.settle + args.steps + k)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(args.top)
