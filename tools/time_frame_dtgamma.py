"""ms per fused 800x800 frame for a few dt_gamma (the bench uses 0; torch-ngp's default for real scenes is 1/128): python tools/time_frame_dtgamma.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
from ngp import workload as W
from ngp.field import NGPFieldFF
from ngp.render import NGPRenderer
dev = torch.device("cuda:0")
field = NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0))
ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
ren.load_density_grid(W.density_grid())
o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(800, 800), 800, 800)
o, d = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
for g in (0.0, 1.0 / 256, 1.0 / 128, 1.0 / 32):
    for _ in range(10):
        out = ren.render_fused(o, d, dt_gamma=g, bg_color=1, image_width=800)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        out = ren.render_fused(o, d, dt_gamma=g, bg_color=1, image_width=800)
    b.record()
    torch.cuda.synchronize()
    n = int(out["stats"][0])
    print(f"dt_gamma {g:.5f}: {a.elapsed_time(b) / 50:.3f} ms per frame, {n / 640000:.1f} samples per ray, {n / (a.elapsed_time(b) / 50) / 1e6:.2f} G samples/s")
