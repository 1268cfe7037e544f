"""Time the table-gradient scatter of a training step on a realistic batch: the march-ordered samples of 4,096 rays (early phase: full
occupancy grid, ~2 M points; --sparse: the S-ring grid, ~0.2 M points).  Atomic kernel (zero fill + k_grid_backward) vs the binned scatter.
   python tools/time_scatter.py [--rays 4096] [--iters 20] [--sparse]"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
import ngp_hip as hip  # noqa: E402
import raymarching  # noqa: E402
from gridencoder import grid as G  # noqa: E402
from ngp import workload as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--sparse", action="store_true")
ap.add_argument("--only", default="both", choices=["both", "atomic", "binned"])
args = ap.parse_args()
dev = torch.device("cuda:0")
o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(200, 200), 200, 200)
idx = np.random.default_rng(0).integers(0, o.shape[0], args.rays)
o, d = torch.from_numpy(o[idx]).to(dev), torch.from_numpy(d[idx]).to(dev)
aabb = torch.tensor([-W.BOUND] * 3 + [W.BOUND] * 3, dtype=torch.float32, device=dev)
nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.05)
if args.sparse:
    bitfield = torch.from_numpy(W.bitfield_from_grid(W.density_grid())[0]).to(dev)
else:
    bitfield = torch.full((2 * 128 ** 3 // 8,), 255, dtype=torch.uint8, device=dev)
counter = torch.zeros(2, dtype=torch.int32, device=dev)
xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, W.BOUND, bitfield, 2, 128, nears, fars, counter, -1, True, 128, False, 0, 1024)
M = int(counter[0].item())                      # the real points (the wrapper returned the unsliced allocation: mean_count unknown)
xyzs = xyzs[:M].contiguous()
enc = G.GridEncoder(desired_resolution=2048 * W.BOUND).to(dev)
inputs = ((xyzs + W.BOUND) / (2 * W.BOUND)).contiguous()
L = enc.num_levels
grad = (torch.randn(L, M, 2, device=dev) * 1e-2).half()
S = float(np.log2(enc.per_level_scale))
lib = hip.lib()
dummy = torch.empty(1, dtype=torch.float16, device=dev)


def atomic():
    g = torch.zeros(enc.embeddings.shape, dtype=torch.float16, device=dev)
    hip.check(lib.ngp_grid_encode_backward(hip.ptr(grad), hip.ptr(inputs), hip.ptr(g), hip.ptr(enc.offsets), hip.ptr(g), M, 3, 2, L, S, 16, 0,
                                           hip.ptr(dummy), hip.ptr(dummy), 0, 0, hip.F16, hip.stream()))
    return g.float()


def binned():
    return G.table_gradient_binned(grad, inputs, enc.offsets, M, L, S, 16, 0, False)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / args.iters


print(f"{args.rays} rays, {M} points")
if args.only in ("both", "atomic"):
    print(f"atomic (zero fill + k_grid_backward + widen): {timed(atomic):.3f} ms")
if args.only in ("both", "binned"):
    print(f"binned (k_gs_bin + k_gs_accumulate, float32 out): {timed(binned):.3f} ms")
if args.only == "both":
    a, b = atomic(), binned()
    print("max |binned - atomic| / max |atomic| =", float((a - b).abs().max() / a.abs().max()))
