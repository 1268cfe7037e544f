"""Throughput of the fused field kernel (ngp_field_forward) and of the drop-in op chain on random points."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")
from ngp import workload as W
from ngp.field import NGPFieldFF
dev = torch.device("cuda:0")
field = NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0)).eval()
for M in (1 << 16, 1 << 20, 1 << 22):
    x = (torch.rand(M, 3, device=dev) * 2 - 1) * W.BOUND
    d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=-1)
    def run(fn, n=10):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    tf = run(lambda: field.forward_fused(x, d))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        tp = run(lambda: field(x, d))
    print("M = %8d   fused field %.3f ms (%.2f G points/s)   op chain %.3f ms (%.2f G points/s)" % (M, tf * 1e3, M / tf / 1e9, tp * 1e3, M / tp / 1e9))
