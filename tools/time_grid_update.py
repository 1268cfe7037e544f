"""Time NGPRenderer.update_extra_state (full sweep and partial refresh) on cuda:0."""
import importlib, os, sys, time
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
import contextlib
from gridencoder import grid as G
for lm in (os.environ.get("NGP_REFRESH_ROWS", "0") != "1",):
  if not lm:
      G.level_major_forward = contextlib.nullcontext            # A/B (NGP_REFRESH_ROWS=1): the row kernel for the refresh's density queries
  print("level-major queries" if lm else "row-kernel queries")
  for label, iters in (("full sweep", 3), ("partial", 6)):
    if label == "partial":
        ren.iter_density = 16
    with torch.autocast("cuda", dtype=torch.float16):
        ren.update_extra_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ren.update_extra_state()
        torch.cuda.synchronize()
    print(" ", label, "%.2f ms per update" % ((time.perf_counter() - t0) / iters * 1e3), "mean density %.4f" % ren.mean_density,
          "occupied bits", int(sum(bin(b).count("1") for b in ren.density_bitfield.cpu().numpy().tobytes()[:4096])))
