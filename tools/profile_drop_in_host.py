"""Where the HOST spends a frame of the unchanged-caller loop (run_cuda, inference branch: 66 iterations, one synchronisation each): per iteration the GPU
idles ~53 us between the count's read-back and the next march (profiles/r16_dropin_summary.md); this splits that gap into the read-back itself and the Python between it
and the next launch.  cProfile over N frames + a hand-timed account of one iteration's calls.
   python tools/profile_drop_in_host.py [--frames 5]"""
import argparse, cProfile, gc, importlib, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
from ngp import workload as W
from ngp.field import NGPFieldFF
from ngp.render import NGPRenderer
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--top", type=int, default=40)
args = ap.parse_args()
dev = torch.device("cuda:0")
ren = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0)), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
ren.load_density_grid(W.density_grid())
o, d = W.get_rays(W.orbit_pose(0, 8), W.intrinsics(800, 800), 800, 800)
o, d = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]


def frame():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        return ren.run_cuda(o, d, dt_gamma=0, bg_color=1, perturb=False, max_steps=1024)


for _ in range(3):
    frame()
torch.cuda.synchronize()
gc.collect(); gc.disable()
t0 = time.perf_counter()
for _ in range(args.frames):
    frame()
torch.cuda.synchronize()
print(f"unprofiled: {1e3 * (time.perf_counter() - t0) / args.frames:.2f} ms per frame")
pr = cProfile.Profile()
pr.enable()
for _ in range(args.frames):
    frame()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(args.top)
