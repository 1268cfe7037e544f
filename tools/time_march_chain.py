"""What one inference-march launch costs when every wave holds a ray that crosses empty space: time k_march_rays for a few ray counts on (a) an empty grid
(every ray crosses the whole box without a sample), (b) the S-ring grid from `near`, (c) a full grid (a sample at once)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
import raymarching
from ngp import workload as W
dev = torch.device("cuda:0")
o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(800, 800), 800, 800)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
aabb = torch.tensor([-W.BOUND] * 3 + [W.BOUND] * 3, dtype=torch.float32, device=dev)
nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.05)
grids = {"empty": torch.zeros(2 * 128 ** 3 // 8, dtype=torch.uint8, device=dev),
         "s-ring": raymarching.packbits(torch.from_numpy(W.density_grid()).to(dev), 10.0),
         "full": torch.full((2 * 128 ** 3 // 8,), 255, dtype=torch.uint8, device=dev)}
for name, bf in grids.items():
    for n in (64, 4096, 65536, 640000):
        alive = torch.arange(n, dtype=torch.int32, device=dev) * (640000 // n)
        rt = nears.clone()
        def run():
            return raymarching.march_rays(n, 1, alive, rt, o, d, W.BOUND, bf, 2, 128, nears, fars, 128, False, 0.0, 1024)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record()
        torch.cuda.synchronize()
        print(f"{name:7s} {n:7d} rays: {a.elapsed_time(b) / 20 * 1e3:8.1f} us per call")
