"""Render one fused 800x800 S-ring frame and the field outputs of 1M ray-ordered points with the library NGP_HIP_LIB selects;
save them under gpurun_out/ for a bit-level comparison between build variants:  python tools/dump_frame.py <tag>"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
from ngp import workload as W
from ngp.field import NGPFieldFF
from ngp.render import NGPRenderer
tag = sys.argv[1]
dev = torch.device("cuda:0")
model = W.make_model(0)
field = NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(model)
ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
ren.load_density_grid(W.density_grid())
o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(800, 800), 800, 800)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
out = ren.render_fused(o[None], d[None], bg_color=1, image_width=800)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
torch.save({k: v.cpu() for k, v in out.items() if k in ("image", "depth", "weights_sum", "stats")}, os.path.join(ROOT, "gpurun_out", f"frame_{tag}.pt"))
print(tag, "saved", out["stats"].tolist())
