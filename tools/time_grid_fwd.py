"""Time the fp32 grid encoder forward on ray-ordered points (the nav filter's 1024 rays x 512 steps), with and without
dy_dx, and its input backward."""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
from ngp import workload as W
from gridencoder import GridEncoder
dev = torch.device("cuda:0")
enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048 * W.BOUND).to(dev)
with torch.no_grad():
    enc.embeddings.uniform_(-0.5, 0.5)
o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(32, 32), 32, 32)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
z = torch.linspace(0.2, 4.0, 512, device=dev)
x = (o[:, None, :] + d[:, None, :] * z[None, :, None]).reshape(-1, 3).clamp(-W.BOUND, W.BOUND).contiguous()
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for p in enc.parameters():
    p.requires_grad_(False)
print("points", x.shape[0])
print("forward, no dy_dx: %.3f ms" % timeit(lambda: enc(x, bound=W.BOUND)))
xg = x.clone().requires_grad_(True)
print("forward with dy_dx: %.3f ms" % timeit(lambda: enc(xg, bound=W.BOUND)))
def fb():
    xg.grad = None
    enc(xg, bound=W.BOUND).sum().backward()
print("forward + input backward (frozen table): %.3f ms" % timeit(fb))
xr = (torch.rand_like(x) * 2 - 1) * W.BOUND
print("forward, no dy_dx, random points: %.3f ms" % timeit(lambda: enc(xr, bound=W.BOUND)))
