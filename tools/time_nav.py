"""Time BASELINE config 4 (nav loop pieces, fp32, no autocast) on cuda:0:
   (ii) density + backward on [20,500,3] body points (planner);  (iii) run() render + backward, 1024 rays x 512 steps (filter)."""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
from ngp import workload as W
from ngp.field import NGPField
from ngp.render import NGPRenderer
dev = torch.device("cuda:0")
torch.manual_seed(0)
field = NGPField(bound=W.BOUND).to(dev)
with torch.no_grad():
    field.encoder.embeddings.uniform_(-0.5, 0.5)
ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=False).to(dev).eval()
rot = torch.tensor([[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]], device=dev)

def timeit(fn, n):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

pts = (torch.rand(20, 500, 3, device=dev) * 2 - 1)
def planner():
    p = pts.clone().requires_grad_(True)
    ren.density(p.reshape(-1, 3) @ rot)["sigma"].sum().backward()
print("(ii) density+backward on 10,000 points: %.3f ms" % timeit(planner, 50))

o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(32, 32), 32, 32)
o, d = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
def filt():
    ro, rd = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    out = ren.render(ro, rd, staged=True, bg_color=1.0, perturb=False, num_steps=512, upsample_steps=0, max_ray_batch=4096)
    out["image"].sum().backward()
print("(iii) run() 1024 rays x 512 steps + backward: %.3f ms" % timeit(filt, 20))

from ngp import nav
q = nav.NavQueries(ren, W.intrinsics(32, 32), 32, 32)
def planner_frozen():
    p = pts.clone().requires_grad_(True)
    q.density_fn(p).sum().backward()
def filt_frozen():
    ro, rd = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    q.render_fn(ro, rd)["image"].sum().backward()
print("frozen model: (ii) %.3f ms   (iii) %.3f ms" % (timeit(planner_frozen, 50), timeit(filt_frozen, 20)))
filt = filt_frozen
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        filt()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
dens = nav.GraphedDensity(q, n_points=pts.numel() // 3)
def planner_graphed():
    p = pts.clone().requires_grad_(True)
    dens(p).sum().backward()
print("frozen model, one graph replay per planner query: (ii) %.3f ms" % timeit(planner_graphed, 200))
