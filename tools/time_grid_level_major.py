"""Half grid forward on the early training batch (ray-ordered points of 4,096 rays through a fully occupied grid): the row kernel (a workgroup walks
all L levels: 16 levels' tables live at once) against the level-major kernel (blockIdx.y = level: one level's 2 MB at a time)."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
import raymarching
import gridencoder.grid as G
from ngp import workload as W
from gridencoder import GridEncoder
dev = torch.device("cuda:0")
enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048 * W.BOUND).to(dev)
with torch.no_grad():
    enc.embeddings.uniform_(-0.5, 0.5)
for p in enc.parameters():
    p.requires_grad_(False)
o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(200, 200), 200, 200)
idx = np.random.default_rng(0).integers(0, o.shape[0], 4096)
o, d = torch.from_numpy(o[idx]).to(dev), torch.from_numpy(d[idx]).to(dev)
aabb = torch.tensor([-W.BOUND] * 3 + [W.BOUND] * 3, dtype=torch.float32, device=dev)
nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.05)
bitfield = torch.full((2 * 128 ** 3 // 8,), 255, dtype=torch.uint8, device=dev)
counter = torch.zeros(2, dtype=torch.int32, device=dev)
xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, W.BOUND, bitfield, 2, 128, nears, fars, counter, -1, True, 128, False, 0, 512)
print("points", xyzs.shape[0])
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def run():
    with torch.autocast("cuda", dtype=torch.float16):
        return enc(xyzs, bound=W.BOUND)
for rows in (True, False):
    G.ROWS_FORWARD = rows
    print("rows kernel" if rows else "level-major + permute", "%.3f ms" % timed(run))
# the same comparison on other point sets: uniformly random points (what the occupancy-grid refresh queries: 2.1 M per refresh) and the first
# iteration of an 800x800 inference loop (640 k march-ordered points, one per ray: neighbouring rays, coherent)
torch.manual_seed(0)
sets = {"random 2.1 M": (torch.rand(2_097_152, 3, device=dev) * 2 - 1) * W.BOUND}
o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(800, 800), 800, 800)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
sets["inference iteration 640 k"] = (o + d * 1.2).clamp(-W.BOUND, W.BOUND).contiguous()
sets["inference iteration 80 k"] = sets["inference iteration 640 k"][::8].contiguous()
for name, pts in sets.items():
    def run_pts():
        with torch.autocast("cuda", dtype=torch.float16):
            return enc(pts, bound=W.BOUND)
    for rows in (True, False):
        G.ROWS_FORWARD = rows
        print(name, "| rows kernel" if rows else "| level-major + permute", "%.3f ms" % timed(run_pts))
