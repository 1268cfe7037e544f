"""Why is the first timed launch of bench.py's timed region ~30 % slower than the rest (BENCH_r03: slowest_launch_index 0, 4.43 against 3.40 ms)?
Hypotheses: the clock ramp after the idle gap between warm-up and the timed region (sync_all + gc.collect + gc.freeze),
the workspace's first touch, k_build_coarse.  This renders the bench's frame after idle gaps of several lengths, each followed by three
back-to-back launches, all between HIP events; a gap of zero with a fresh workspace separates first touch from idleness.
   python tools/first_launch.py"""
import gc
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
from ngp import workload as W  # noqa: E402
from ngp.field import NGPFieldFF  # noqa: E402
from ngp.render import NGPRenderer  # noqa: E402

dev = torch.device("cuda:0")
field = NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0))
ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
ren.load_density_grid(W.density_grid())
o, d = W.get_rays(W.orbit_pose(0, 8), W.intrinsics(800, 800), 800, 800)
o, d = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]


def frame():
    return ren.render_fused(o, d, dt_gamma=0, bg_color=1, max_steps=1024, image_width=800)


def burst(n=3):
    ev = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); frame(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    return [round(a.elapsed_time(b), 3) for a, b in ev]


for _ in range(20):
    frame()
torch.cuda.synchronize()
print("steady (no gap):", burst(5))
for gap_ms in (0, 1, 5, 20, 50, 100, 300, 1000):
    torch.cuda.synchronize()
    time.sleep(gap_ms * 1e-3)
    print(f"after {gap_ms:5d} ms idle:", burst(4))
t0 = time.perf_counter(); gc.collect(); gc.freeze(); t1 = time.perf_counter()
print(f"gc.collect + freeze took {1e3 * (t1 - t0):.1f} ms of host time; then:", burst(4))
gc.unfreeze()
# a busy GPU during the host-side gap (a stream of tiny kernels keeps the clocks up): same gap, no ramp?
x = torch.zeros(1 << 20, device=dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1:
    x.add_(1.0)
print("after 100 ms of tiny kernels instead of idleness:", burst(4))
