"""Entries per (level, slice) of the binned scatter on the tools/time_scatter.py batch (reads the directory k_gs_bin leaves in the workspace)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
import ngp_hip as hip, raymarching
from gridencoder import grid as G
from ngp import workload as W
dev = torch.device("cuda:0")
o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(200, 200), 200, 200)
idx = np.random.default_rng(0).integers(0, o.shape[0], 4096)
o, d = torch.from_numpy(o[idx]).to(dev), torch.from_numpy(d[idx]).to(dev)
aabb = torch.tensor([-W.BOUND] * 3 + [W.BOUND] * 3, dtype=torch.float32, device=dev)
nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.05)
sparse = "--sparse" in sys.argv
bitfield = torch.from_numpy(W.bitfield_from_grid(W.density_grid())[0]).to(dev) if sparse else torch.full((2 * 128 ** 3 // 8,), 255, dtype=torch.uint8, device=dev)
counter = torch.zeros(2, dtype=torch.int32, device=dev)
xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, W.BOUND, bitfield, 2, 128, nears, fars, counter, -1, True, 128, False, 0, 1024)
M = int(counter[0].item())
xyzs = xyzs[:M].contiguous()
enc = G.GridEncoder(desired_resolution=2048 * W.BOUND).to(dev)
inputs = ((xyzs + W.BOUND) / (2 * W.BOUND)).contiguous()
L = 16
grad = (torch.randn(L, M, 2, device=dev) * 1e-2).half()
lib = hip.lib()
rows, n, _ = G.offsets_info(enc.offsets)
out = torch.empty(n, 2, device=dev)
ws = hip.workspace(lib.ngp_grid_scatter_binned_workspace(M, L), dev)
hip.check(lib.ngp_grid_scatter_binned(hip.ptr(grad), hip.ptr(inputs), hip.ptr(enc.offsets), hip.ptr(out), M, L, float(np.log2(enc.per_level_scale)), 16, rows, 0, 0, hip.F32, 1.0,
                                      hip.ptr(ws), ws.numel(), hip.stream()))
torch.cuda.synchronize()
nchunks = (min(M, 1 << 22) + 1023) // 1024
dirs_ = ws[256:256 + L * 129 * nchunks * 2].view(torch.int16).cpu().numpy().view(np.uint16).reshape(L, 129, nchunks).astype(np.int64)
per = (dirs_[:, 1:, :] - dirs_[:, :-1, :]).sum(axis=2)          # [L, 128]
print("points", M, "regions", nchunks, "entries", int(per.sum()), "per point", per.sum() / M)
for l in range(L):
    nz = per[l][per[l] > 0]
    print(f"level {l:2d}: slices {len(nz):3d} entries {int(nz.sum()):9d}  max/slice {int(nz.max()):8d}  mean/slice {int(nz.mean()):8d}")
