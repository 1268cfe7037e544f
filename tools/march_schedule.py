"""Per-iteration cost of the drop-in inference loop's march: (n_alive, n_step, samples found, ms of the march call) for one 800x800 S-ring frame."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
import ngp_hip
from ngp import workload as W
from ngp.field import NGPFieldFF
from ngp.render import NGPRenderer
dev = torch.device("cuda:0")
field = NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0))
ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
ren.load_density_grid(W.density_grid())
o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(800, 800), 800, 800)
o, d = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
for rep in range(2):
    trace = []
    ngp_hip.TIMERS = {}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        ren.run_cuda(o, d, bg_color=1, trace=trace)
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ngp_hip.TIMERS["march_rays"]]
    ngp_hip.TIMERS = None
print("iter  n_alive  n_step  samples  march_ms")
for k, ((na, ns, sm), t) in enumerate(zip(trace, ms)):
    print(f"{k:3d} {na:8d} {ns:3d} {sm:9d} {t:8.3f}")
print("total march ms", sum(ms), "iterations", len(ms))
