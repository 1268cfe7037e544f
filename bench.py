#!/usr/bin/env python3
"""Headline benchmark: ray-samples/s of 800x800 novel-view renders of the synthetic Stonehenge stand-in
("S-ring", SURVEY.md 8d / BASELINE config 2) through the fused gfx950 path (csrc/render_fused.hip).

  python bench.py --gpus N --steps K --warmup W
N > 1 runs one rank per GPU.  Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or -- `python bench.py --gpus N` with WORLD_SIZE unset -- this process
starts them itself (`launch_ranks`: a child `torch.distributed.run --standalone`-style launch on 127.0.0.1, BEFORE anything here touches a GPU),
relays rank 0's one JSON line and exits with the children's status.  A step = one frame = ONE kernel launch over 640,000 rays already resident in HBM.  Rays shard
embarrassingly: every rank renders its own stream of camera poses with a replica of the model (25 MB half table +
36 KB of weights + 0.5 MB bitfield) and there is no collective on the data path -> weak scaling; the only
collectives are the timing barrier and the reduction of the counters.

Prints ONE JSON line (rank 0).  `roofline` prices the render kernel against HBM with the algorithmic gather bytes
(512 B per ray-sample); `cpu_baseline` is the CPU oracle (a port: the reference has no CPU path) on a bounded sample.
"""
import argparse
import contextlib
import gc
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

GATHER_BYTES_PER_SAMPLE = 512          # 16 levels x 8 corners x 2 features x 2 B
MLP_FLOPS_PER_SAMPLE = 2 * (64 * (32 + 64 + 16) + 64 * (32 + 128 + 16))   # FFMLP shapes of nerf/network_ff.py
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
MFMA_PEAK_TFLOPS = 2500.0              # dense f16 MFMA peak (MI355X_MICROARCH.md)
LANE_ADDRESSES_PER_SAMPLE = 11 * 8 + 5 * 4   # per-lane gather addresses the texture path sees per ray-sample
# distinct 64-byte lines one ray-sample touches: dense levels 0-4 read 4 x-pairs each (4 lines), hashed levels 5-15 read 8 entries
# whose x-neighbours share a line 15 times out of 16 (4 + 1/4 lines): 5*4 + 11*4.25 = 66.75  (profiles/r13: 58-69 measured)
LINES_PER_SAMPLE = 5 * 4 + 11 * 4.25


MODEL_NAMES = {"handset": "hand-set (table and weights set by hand to represent the scene; bit-reproducible)",
               "trained": "fitted (a fresh field trained on renders of the hand-set scene with the package's trainer, seed 0)"}
ATOMIC_PEAK_GBS = 1300.0               # MI355X_MICROARCH.md "Global float atomics": ~1.3 TB/s of added bytes, chip-wide
SCATTER_BYTES_PER_POINT = 512          # 16 levels x 8 corners x 2 features x 2 B of half2 atomics (SURVEY 8d "training extra")


FRAME_SOURCES = ("render_fused.hip", "ngp_field.h", "ngp_device.h", "ngp_march.h", "ngp_mlp.h", "ngp_sh.h", "ngp_camera.h", "Makefile")   # what k_render_frame_multi is built from
TRAIN_SOURCES = ("gridencoder.hip", "ngp_device.h", "Makefile")                                                              # ... and k_grid_backward


@contextlib.contextmanager
def no_gc_pauses():
    """Timed regions run with Python's cyclic garbage collector off.  A generation-2 collection of this process (torch + numpy: millions of tracked
    objects) takes ~45 ms of HOST time; the HIP runtime lets the host run only a few launches ahead of the GPU, so the pause starves the queue and shows up
    as ONE 45 ms step among 3.3 ms ones -- always at the same step of a run (measured: launch 76 of 100, 3.76 against 3.32 ms per step on average).  It is
    host bookkeeping, not part of the path being measured; a long-running viewer would call gc.freeze() after set-up for the same reason."""
    gc.collect()
    gc.freeze()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()
        gc.unfreeze()


def sources_sha16(files=FRAME_SOURCES):
    """hash of the kernel sources a profiled kernel is built from: profiles/<tag>_meta.json records it, so a PMC summary collected for
    an older kernel cannot pass as this one's traffic"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "nerf-navigation_amd", "csrc")
    for name in files:
        data = open(os.path.join(csrc, name), "rb").read()
        if name == "Makefile":                         # only what reaches the compiler: tool, architecture, flags (not the list of source files)
            keep, cont = [], False
            for line in data.decode().splitlines():
                if cont or line.startswith(("HIPCC", "ARCH", "HIPFLAGS")):
                    keep.append(line)
                    cont = line.rstrip().endswith("\\")
            data = "\n".join(keep).encode()
        h.update(data)
    return h.hexdigest()[:16]


def committed_counters(pattern, files):
    """per-dispatch PMC means of the newest committed profile matching `pattern`, or (None, reason) when it was collected for other sources"""
    import csv
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not found:
        return None, {"reason": "no committed profile"}
    pmc = found[-1]
    meta = pmc.replace("_pmc.csv", "_meta.json")
    recorded = json.load(open(meta)).get("sources_sha16") if os.path.exists(meta) else None
    if recorded != sources_sha16(files):
        return None, {"profile": os.path.relpath(pmc, ROOT), "profile_sources": recorded, "built_sources": sources_sha16(files)}
    return {r["counter"]: float(r["mean_per_dispatch"]) for r in csv.DictReader(open(pmc))}, {"profile": os.path.relpath(pmc, ROOT)}


def bench_train(args, rank, world, dev, W, teacher):
    """Secondary metric: training throughput of the op-by-op path with autograd (main_nerf.py's recipe, ngp/train.py).
    Each rank draws its own 4096-ray batches from its own views; gradients are averaged with one all-reduce per step.
    Two phases are timed: `warmup_phase` = the first steps on the initial (full) occupancy grid, and the headline `steady` phase
    after args.settle steps, when the grid has converged to the scene and a step marches ~10x fewer points."""
    import torch.distributed as dist
    import ngp_hip
    from ngp import sharding
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    from ngp.train import NGPTrainer
    res, n_rays = 200, 4096
    intr = W.intrinsics(res, res)
    radius, height = W.scene_orbit(args.workload)
    pool = []
    for view in sharding.pose_indices(rank, world, 8):
        o, d = W.get_rays(W.orbit_pose(view, 8 * world, radius, height), intr, res, res)
        to, td = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
        pool.append((to, td, teacher.render_fused(to, td, bg_color=1, image_width=res)["image"]))
    torch.manual_seed(0)                                                    # identical initial replicas
    student = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev)
    tr = NGPTrainer(student, lr=1e-2, iters=30000, fp16=True, steps_per_epoch=len(pool), time_exchange=True)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)

    def step(k):
        to, td, tc = pool[k % len(pool)]
        idx = torch.randint(0, res * res, (n_rays,), device=dev, generator=gen)
        return tr.step(to[:, idx], td[:, idx], tc[:, idx], bg_color=1, max_steps=1024)

    def sync_all():
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()

    def points_now():
        n = min(16, student.local_step)
        return float(student.step_counter[:n, 0].float().mean().item()) if n else float(student.mean_count)

    host_ms = [None]                                                       # ms per step the host needed to queue the last timed phase

    def timed_phase(first, count):
        ngp_hip.TIMERS = {}
        with no_gc_pauses():
            sync_all()
            t0 = time.perf_counter()
            for k in range(count):
                loss = step(first + k)
            queued = time.perf_counter() - t0                                  # the host has queued every launch (an upper bound of the host's need: the runtime lets it run only a few launches ahead)
            sync_all()
            elapsed = time.perf_counter() - t0
        scatter_ms, calls = ngp_hip.timer_ms("grid_encode_backward")
        ngp_hip.TIMERS = None
        host_ms[0] = queued / count * 1e3
        return elapsed, float(loss), points_now(), scatter_ms, calls

    k0 = 0
    for k in range(args.warmup):
        step(k0 + k)
    k0 += args.warmup
    warm = timed_phase(k0, args.steps)                                      # early phase: full occupancy grid
    k0 += args.steps
    for k in range(args.settle):
        step(k0 + k)
    k0 += args.settle
    elapsed, loss, points, scatter_ms, calls = timed_phase(k0, args.steps)  # steady state
    rays_all, t_max = sharding.reduce_throughput(n_rays * args.steps, elapsed, dev)
    if rank == 0:
        def roof(pts, ms, profiled_phase=False):
            """the table-gradient scatter (binned: k_gs_bin + k_gs_accumulate, event-timed together): 512 algorithmic bytes of contributions per
            point against HBM (no atomics left: the bound is the entry traffic), with the float-atomic peak the replaced kernel was priced at beside it"""
            if not ms:
                return None
            a = SCATTER_BYTES_PER_POINT * pts / (ms * 1e-3) / 1e9
            r = {"bound": "hbm", "kernel": "k_gs_bin + k_gs_accumulate (binned scatter, csrc/gridencoder.hip)", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": a / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ms,
                 "algorithmic_bytes_per_point": SCATTER_BYTES_PER_POINT, "points_per_launch": pts,
                 "vs_float_atomic_peak": a / ATOMIC_PEAK_GBS, "float_atomic_peak_GBs": ATOMIC_PEAK_GBS,
                 "replaced": "k_grid_backward<half,3,2>: 2.68 ms at 2.0 M points (profiles/r14_train), 0.29-0.39 of the atomic peak"}
            c, info = committed_counters("r[0-9][0-9]_train_pmc.csv", TRAIN_SOURCES)
            if c is not None and c.get("WRITE_SIZE") and profiled_phase:   # the committed profile was collected on the early grid: the warm-up phase's launches
                r["profile"] = info["profile"]
                r["traffic"] = (c["WRITE_SIZE"] + c.get("FETCH_SIZE", 0.0)) * 1024.0
                r["traffic_unit"] = "bytes per launch of k_gs_bin alone, fabric side (profiles/*_train_pmc.csv lists the first-named kernel)"
            return r
        print(json.dumps({
            "metric": "training rays/sec (4096-ray steps, FFMLP field under autocast, Adam, grid refresh every 16 steps)",
            "value": rays_all / t_max, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"S-{args.workload} training (teacher-rendered 200x200 views), native field forward + backward, march, composite and table scatter under autograd, steady state after "
                                   f"{args.warmup + args.steps + args.settle} steps",
                       "rays_per_step_per_gpu": n_rays, "points_per_step": points, "final_loss": loss,
                       "collectives": ("nccl (RCCL)" if dist.is_initialized() and dist.get_backend() == "nccl" else ("gloo rehearsal" if dist.is_initialized() else "none (one process)"))
                                      + (", forced on a single rank" if args.force_dist and world == 1 else "")},
            # gradient exchange of the LAST step: bytes handed to all-reduce (half table + float32 weight bucket on the native route) and the GPU time
            # between the end of the backward and the gradients being ready (events on the compute stream); zero / null without a process group
            "allreduce_bytes": tr.exchange.stats["allreduce_bytes"], "exposed_collective_ms": tr.exchange.exposed_ms(),
            "weight_ema": {"decay": 0.95, "updates": tr.ema.num_updates, "every_steps": len(pool)},
            "host_queue_ms_per_step": host_ms[0],
            "roofline": roof(points, scatter_ms),
            "warmup_phase": {"ms_per_step": 1e3 * warm[0] / args.steps, "points_per_step": warm[2], "loss": warm[1],
                             "roofline": roof(warm[2], warm[3], profiled_phase=True)}}))
    if dist.is_initialized():
        dist.destroy_process_group()


def fit_model(args, dev, W, teacher):
    """"table/MLP obtained by fitting the field with the build's own trainer for a fixed 2,000 steps, seed 0".  The teacher is the
    hand-set model (it IS the scene); the student starts from the reference's initialisation and its own, learned, occupancy grid."""
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer
    from ngp.train import NGPTrainer
    res, n_rays = 200, 4096
    intr = W.intrinsics(res, res)
    radius, height = W.scene_orbit(args.workload)
    pool = []
    for view in range(24):
        o, d = W.get_rays(W.orbit_pose(view, 24, radius, height + 0.25 * ((view % 3) - 1)), intr, res, res)
        to, td = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
        pool.append((to, td, teacher.render_fused(to, td, bg_color=1, image_width=res)["image"]))
    torch.manual_seed(0)
    student = NGPRenderer(NGPFieldFF(bound=W.BOUND).to(dev), bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev)
    tr = NGPTrainer(student, lr=1e-2, iters=args.fit_steps, fp16=True, steps_per_epoch=len(pool))       # 24 views = one epoch: the weight average follows
    gen = torch.Generator(device=dev).manual_seed(1)

    def step(k):
        to, td, tc = pool[k % len(pool)]
        idx = torch.randint(0, res * res, (n_rays,), device=dev, generator=gen)
        return tr.step(to[:, idx], td[:, idx], tc[:, idx], bg_color=1, max_steps=1024)

    # The LAST `train_steps` steps of the fit are the `train` block of the line (BASELINE config 3 in its steady state: the occupancy grid has
    # converged, a step marches ~0.6 M points): wall time between two synchronisations, and the field's and the table scatter's launches between
    # HIP events on the launch stream (ngp_hip.timed).  The same 2,000 steps as without the block: nothing is added to the fit.
    import ngp_hip
    tail = 0 if args.no_train_block else max(0, min(int(args.train_steps), args.fit_steps // 2))
    t0 = time.perf_counter()
    # ... and steps [32, 32 + tail) are its early phase (the occupancy grid still nearly full: ~1.8 M points per step): two more synchronisations, no more steps
    early = None
    head0 = 32 if tail and args.fit_steps - tail >= 32 + tail else args.fit_steps
    if head0 == args.fit_steps:
        for k in range(args.fit_steps - tail):
            loss = step(k)
    else:
        with no_gc_pauses():                               # (entered before the untimed steps: the collector's own 36 ms must not idle the GPU just before the timed ones)
            for k in range(head0):
                loss = step(k)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(head0, head0 + tail):
                loss = step(k)
            torch.cuda.synchronize()
            early_wall = time.perf_counter() - t1
        n = min(16, student.local_step)
        early = {"steps": tail, "first_step": head0, "ms_per_step": 1e3 * early_wall / tail, "rays_per_s": n_rays * tail / early_wall,
                 "points_per_step": float(student.step_counter[:n, 0].float().mean().item())}
        for k in range(head0 + tail, args.fit_steps - tail):
            loss = step(k)
    train = None
    if tail:
        ngp_hip.TIMERS = {}
        with no_gc_pauses():
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(args.fit_steps - tail, args.fit_steps):
                loss = step(k)
            queued = time.perf_counter() - t1
            torch.cuda.synchronize()
            wall = time.perf_counter() - t1
        timers = {name: ngp_hip.timer_ms(name)[0] for name in ("field_train_forward", "field_train_backward", "grid_encode_backward", "adam_step")}
        ngp_hip.TIMERS = None
        n = min(16, student.local_step)
        points = float(student.step_counter[:n, 0].float().mean().item())

        def roof(kernel, ms):
            if not ms:
                return None
            a = SCATTER_BYTES_PER_POINT * points / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": kernel, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS, "avg_launch_ms": ms,
                    "algorithmic_bytes_per_point": SCATTER_BYTES_PER_POINT, "points_per_launch": points}
        train = {"what": f"the last {tail} of the {args.fit_steps} fit steps (4,096 rays per step, FFMLP field under autocast, native Adam + GradScaler, grid refresh every 16 steps: "
                         "BASELINE config 3, steady state)", "steps": tail, "ms_per_step": 1e3 * wall / tail, "rays_per_s": n_rays * tail / wall,
                 "points_per_step": points, "host_queue_ms_per_step": 1e3 * queued / tail,
                 "forward_roofline": roof("k_ft_encode_levels + k_field_train_forward (level-by-level gather of 512 B per point, then both networks), event-timed together", timers["field_train_forward"]),
                 "scatter_roofline": roof("k_gs_bin + k_gs_accumulate (binned table-gradient scatter)", timers["grid_encode_backward"]),
                 "field_backward_ms": timers["field_train_backward"], "optimizer_ms": timers["adam_step"],
                 "early_phase": early}
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    student.eval()
    to, td, tc = pool[1]
    with torch.no_grad():
        live = float(-10 * torch.log10(torch.mean((student.render_fused(to, td, bg_color=1, image_width=res)["image"] - tc) ** 2)))
    # the reference evaluates (and saves its best checkpoint) with the exponential moving average of the weights (nerf/utils.py:851-853,
    # :983-994): install it for everything that follows -- the timed renders and the PSNR legs of the bench
    tr.ema.store()
    tr.ema.copy_to()
    student.field.mark_updated()
    with torch.no_grad():
        img = student.render_fused(to, td, bg_color=1, image_width=res)["image"]
        psnr = float(-10 * torch.log10(torch.mean((img - tc) ** 2)))
    return student, train, {"steps": args.fit_steps, "seconds": seconds, "final_loss": float(loss), "psnr_vs_teacher_db": psnr,
                     "weights": f"exponential moving average (decay 0.95, {tr.ema.num_updates} epoch updates), as the reference evaluates", "psnr_live_weights_db": live,
                     "occupied_cells": int(torch.count_nonzero(student.density_grid > min(student.mean_density, 10.0)))}


def psnr_ref(pred, truth):
    """PSNRMeter.update's formula (nerf/utils.py:203-210): -10 log10(mean((pred - truth)^2)), peak value 1"""
    return float(-10 * np.log10(np.mean((np.asarray(pred, np.float32) - np.asarray(truth, np.float32)) ** 2)))


def fitted_block(args, dev, W, teacher, rays, Wd):
    """SURVEY 8d cfg 2 as contracted: the SAME frame kernel on a model fitted for --fit-steps steps (seed 0) with the package's own trainer --
    untimed fit, then --fit-frames timed 800x800 launches with HIP events -- plus the north_star's PSNR criterion on a <= 200^2 view:
    psnr_delta_db = PSNR(HIP render of the student, teacher) - PSNR(CPU-oracle render of the same student, teacher)."""
    from oracle import ngp_oracle as O, render_oracle as R
    student, train, fit = fit_model(args, dev, W, teacher)
    N = rays[0][0].shape[1]
    ev, stats = [], []
    with no_gc_pauses():
        for k in range(5):
            student.render_fused(*rays[k % len(rays)], dt_gamma=0, bg_color=1, max_steps=1024, image_width=Wd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.fit_frames):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = student.render_fused(*rays[k % len(rays)], dt_gamma=0, bg_color=1, max_steps=1024, image_width=Wd)
            b.record()
            ev.append((a, b)); stats.append(out["stats"])
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    samples = int(torch.stack(stats)[:, 0].to(torch.int64).sum().item())
    kernel_s = 1e-3 * float(np.mean([a.elapsed_time(b) for a, b in ev]))
    per_launch = samples / args.fit_frames
    achieved = GATHER_BYTES_PER_SAMPLE * per_launch / kernel_s / 1e9
    block = {"model": MODEL_NAMES["trained"], "fit": fit, "train": train, "frames": args.fit_frames, "value": samples / elapsed, "unit": "ray-samples/s",
             "ms_per_step": 1e3 * elapsed / args.fit_frames, "fps": args.fit_frames / elapsed, "samples_per_ray": per_launch / N,
             "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                          "kernel": "k_render_frame_multi", "avg_launch_ms": 1e3 * kernel_s}}
    if not args.no_cpu:
        O.build()
        cpu_threads = int(os.environ.get("NGP_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
        O.set_threads(cpu_threads)
        f = student.field
        sm = dict(embeddings=f.encoder.embeddings.detach().float().cpu().numpy(), offsets=f.encoder.offsets.cpu().numpy(),
                  per_level_scale=float(f.encoder.per_level_scale), sigma_weights=f.sigma_net.weights.detach().float().cpu().numpy(),
                  color_weights=f.color_net.weights.detach().float().cpu().numpy(), bound=W.BOUND)
        r = min(int(args.fit_cpu_res), 200)
        radius, height = W.scene_orbit(args.workload)
        o, d = W.get_rays(W.orbit_pose(3, 8, radius, height), W.intrinsics(r, r), r, r)       # a view between two training views
        to, td = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
        truth = teacher.render_fused(to, td, bg_color=1, image_width=r)["image"][0].cpu().numpy()
        hip = student.render_fused(to, td, bg_color=1, image_width=r)["image"][0].cpu().numpy()
        t1 = time.perf_counter()
        ref = R.run_cuda(lambda x, dd: R.field_forward(sm, x, dd, 1.0), o, d, student.density_bitfield.cpu().numpy(), W.BOUND, 2)
        cpu_s = time.perf_counter() - t1
        p_hip, p_cpu = psnr_ref(hip, truth), psnr_ref(ref["image"], truth)
        block.update({"psnr_view": f"{r}x{r}, orbit pose 3 of 8 (not a training view)", "psnr_hip_vs_teacher_db": p_hip, "psnr_oracle_vs_teacher_db": p_cpu,
                      "psnr_delta_db": p_hip - p_cpu, "psnr_vs_oracle_db": psnr_ref(hip, ref["image"]),
                      "max_abs_vs_oracle": float(np.max(np.abs(hip - ref["image"]))), "oracle_seconds": cpu_s, "oracle_ray_samples": ref["samples"],
                      "criterion": "north_star: PSNR within 0.1 dB of the CPU path", "criterion_met": bool(abs(p_hip - p_cpu) < 0.1)})
    return block


def nav_block(args, dev, W):
    """BASELINE config 4 on the fused float32 kernels (csrc/nav_field.hip, ngp.nav.NativeNavQueries) over the default field holding the S-ring scene:
    (iii) one pose-filter iteration = run() on 1,024 rays x 512 steps + backward to the rays (simulate.py:203-205, nav/estimator_helpers.py:293-327);
    (ii) the planner's query = density + gradient on [20, 500, 3] body points (nav/quad_plot.py:224-250), as one hipGraph replay; (i) the A* occupancy
    query on 100^3 points (nav/quad_plot.py:65-79).  Roofline: 1,024 B of float32 gather per sample against HBM, the forward and the
    (re-gathering) backward launch each event-timed on the launch stream."""
    import ngp_hip
    from ngp import nav
    from ngp.field import NGPField
    from ngp.render import NGPRenderer
    model = W.make_model(0)
    sw, cw = W.nav_weights(0)
    field = NGPField(bound=W.BOUND).to(dev)
    with torch.no_grad():
        field.encoder.embeddings.copy_(torch.from_numpy(model["embeddings"]))
        for layer, w in zip(list(field.sigma_net) + list(field.color_net), sw + cw):
            layer.weight.copy_(torch.from_numpy(w))
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=False).to(dev).eval()
    rays_n, steps_n = 1024, 512
    q = nav.NativeNavQueries(ren, W.intrinsics(32, 32), 32, 32, num_steps=steps_n)
    o, d = W.get_rays(W.orbit_pose(1), W.intrinsics(32, 32), 32, 32)
    o, d = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]

    def filter_iteration():
        ro, rd = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
        q.render_fn(ro, rd)["image"].sum().backward()

    pts = (torch.rand(20, 500, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * 2 - 1)
    graphed = nav.GraphedDensity(nav.NavQueries(ren, W.intrinsics(32, 32), 32, 32), n_points=10000)

    def planner_query():                           # what NativeNavQueries.density_fn does with a planner-sized batch: ONE launch, value and Jacobian
        p = pts.clone().requires_grad_(True)
        q.density_fn(p).sum().backward()

    def planner_query_graphed_chain():             # round 3's route: the level-parallel op chain + its backward as one hipGraph replay
        p = pts.clone().requires_grad_(True)
        graphed(p).sum().backward()

    lin = torch.linspace(-1, 1, 100, device=dev)
    lattice = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), dim=-1)

    def astar_query():
        with torch.no_grad():
            q.density_fn(lattice)

    def timeit(fn, n):
        with no_gc_pauses():                       # (parked before the warm-up: see the headline region)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

    n = int(args.nav_iters)
    planner_ms, planner_graph_ms, astar_ms = timeit(planner_query, 4 * n), timeit(planner_query_graphed_chain, 4 * n), timeit(astar_query, n)
    ngp_hip.TIMERS = {}
    filter_ms = timeit(filter_iteration, n)
    fwd_ms, bwd_ms = ngp_hip.timer_ms("nav_run_forward")[0], ngp_hip.timer_ms("nav_run_backward")[0]
    ngp_hip.TIMERS = None
    samples = rays_n * steps_n

    def roof(kernel, ms):
        a = 1024.0 * samples / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS, "avg_launch_ms": ms,
                "algorithmic_bytes_per_sample": 1024, "samples_per_launch": samples}
    return {"what": "BASELINE config 4 (nav loop queries, float32, default field holding the S-ring scene) on the fused kernels of csrc/nav_field.hip",
            "filter_iteration_ms": filter_ms, "filter_iteration": f"run() {rays_n} rays x {steps_n} steps + backward to the rays", "iterations": n,
            "filter_samples_per_s": samples / (filter_ms * 1e-3),
            "planner_query_ms": planner_ms, "planner_query": "density + gradient on 10,000 body points: one launch of k_nav_density_vj (16 levels over four waves, x @ rot folded in) "
                                                             "+ the torch ops of sum().backward()",
            "planner_query_graphed_chain_ms": planner_graph_ms,
            "astar_query_ms": astar_ms, "astar_query": "density on the 100^3 lattice, no gradient (k_nav_density_fwd)",
            "forward_roofline": roof("k_nav_run_fwd", fwd_ms), "backward_roofline": roof("k_nav_run_bwd (recomputes the gather)", bwd_ms)}


def drop_in_block(args, ren, rays, N):
    """The loop an UNMODIFIED nerf/renderer.py:325-374 runs per frame through the drop-in packages (march_rays -> field -> composite_rays -> compaction,
    one host synchronisation per iteration as there): the path north_star means by "call them unchanged".  800x800, the same rays as the headline."""
    def frame(k):
        o, d = rays[k % len(rays)]
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return ren.run_cuda(o, d, dt_gamma=0, bg_color=1, perturb=False, max_steps=1024)
    n = int(args.drop_in_frames)
    with no_gc_pauses():
        for k in range(2):
            frame(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            frame(k)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / n
    return {"what": "run_cuda as an unmodified renderer runs it (nerf/renderer.py:325-374: 66 iterations of march / field / composite over the drop-in ops, the mask line :365 as raymarching.compact_alive with its count polled by the host), FFMLP field under autocast, same frames as the headline",
            "frames": n, "ms_per_frame": 1e3 * sec, "fps": 1.0 / sec, "rays_per_frame": N}


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as a CHILD process group (torch.distributed.run on the loopback), relay
    rank 0's JSON line, exit with the children's status.  Runs before this process has made any GPU call (importing torch does not initialise the
    device) and never replaces the process (no exec)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or len(lines) != 1:
        print(f"bench.py: the {args.gpus}-rank launch failed (exit {proc.returncode}, {len(lines)} result lines)", file=sys.stderr)
        sys.exit(proc.returncode or 1)
    print(lines[0])
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # ~1 s of GPU time; runs shorter than ~50 frames scatter by +-10 %
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--cpu-res", type=int, default=400, help="side of the frame the CPU oracle renders for cpu_baseline / PSNR (about 10 s of CPU work with a team of 16; BASELINE.md 2 quotes 200x200)")
    ap.add_argument("--model", default="handset", choices=["handset", "trained"],
                    help="handset = the field whose table / weights are set by hand to represent the scene (headline); trained = a fresh field fitted to "
                         "renders of it with the package's own trainer for --fit-steps steps, seed 0 (SURVEY 8d), then rendered")
    ap.add_argument("--fit-steps", type=int, default=2000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-fit", action="store_true", help="skip the `fitted` block (the same frame kernel timed on a model fitted for --fit-steps steps, SURVEY 8d cfg 2)")
    ap.add_argument("--fit-frames", type=int, default=30, help="timed 800x800 frames of the fitted model (after 5 warm-up frames)")
    ap.add_argument("--fit-cpu-res", type=int, default=200, help="side of the view on which the fitted student is rendered by the CPU oracle too (psnr_delta_db); <= 200")
    ap.add_argument("--frames-per-launch", type=int, default=8, help="secondary `multi_frame` block: this many poses rendered by ONE launch (1 = skip)")
    ap.add_argument("--force-dist", action="store_true", help="N = 1: initialise the nccl (= RCCL) process group anyway and take every collective of the N > 1 path")
    ap.add_argument("--path", default="fused", choices=["fused", "fused_camera", "fused_torch_rays", "drop_in", "per_op", "per_op_fused_field"],
                    help="fused = headline (rays resident); fused_camera = rays generated inside the frame kernel from the pose; "
                         "fused_torch_rays = torch get_rays per frame + fused; drop_in = run_cuda as an unmodified renderer runs it (the field picks its one-launch route); "
                         "per_op = the same loop with every op of the field on its own; per_op_fused_field = the one-launch field called explicitly")
    ap.add_argument("--workload", default="ring", choices=["ring", "church"],
                    help="ring = S-ring, the BASELINE config-2 stand-in (headline); church = the larger shell scene of config 5")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank renders its own frames (headline); strong = every frame is cut into row bands, one per rank")
    ap.add_argument("--gather", action="store_true", help="strong scaling: all-gather the row bands into the full image on every rank, inside the timed region")
    ap.add_argument("--settle", type=int, default=1460, help="train mode: untimed steps between the warm-up phase and the steady-state phase")
    ap.add_argument("--mode", default="render", choices=["render", "train"],
                    help="render = the headline metric; train = secondary: training steps (configs 3 / 5), 4096 rays per step per GPU")
    ap.add_argument("--train-steps", type=int, default=64, help="`train` block: the last this-many steps of the fit are timed (0 / --no-train-block = skip)")
    ap.add_argument("--no-train-block", action="store_true")
    ap.add_argument("--nav-iters", type=int, default=50, help="`nav` block: timed pose-filter iterations (BASELINE config 4)")
    ap.add_argument("--no-nav-block", action="store_true")
    ap.add_argument("--drop-in-frames", type=int, default=5, help="`drop_in` block: timed 800x800 frames of the unchanged-caller loop")
    ap.add_argument("--no-drop-in-block", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # NGP_BENCH_REHEARSAL=1: every rank on cuda:0 over gloo -- exercises the N > 1 code path on a one-GPU box (RCCL refuses two
    # ranks on one device); timing lines from a rehearsal are not measurements
    rehearsal = os.environ.get("NGP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1 or args.force_dist:
        if world == 1:                                       # --force-dist: a single-rank RCCL group (no launcher: rendezvous on the loopback)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    use_dist = dist.is_initialized()

    importlib.import_module("nerf-navigation_amd")
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    from ngp.render import NGPRenderer

    model = W.make_model(0, scene=args.workload)
    grid = W.density_grid(scene=args.workload)
    field = NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(model)
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
    ren.load_density_grid(grid)

    if args.mode == "train":
        from ngp import sharding as _sh
        _sh.FORCE_COLLECTIVES = bool(args.force_dist)
        return bench_train(args, rank, world, dev, W, ren)

    fit = None
    if args.model == "trained":
        ren, _, fit = fit_model(args, dev, W, ren)               # untimed: the bench renders the fitted model
        model = None

    H = Wd = args.res
    intr = W.intrinsics(H, Wd)
    n_poses = 8
    radius, height = W.scene_orbit(args.workload)
    from ngp import sharding
    sharding.FORCE_COLLECTIVES = bool(args.force_dist)
    strong = args.scaling == "strong" and world > 1
    rays, poses = [], []
    if strong:
        # strong scaling: ONE stream of frames; rank r renders rows [lo, hi) of every frame (contiguous bands keep ray coherence)
        assert args.path == "fused", "--scaling strong is implemented for the fused path"
        lo, hi = sharding.row_band(rank, world, H)
        for view in range(n_poses):
            poses.append(W.orbit_pose(view, n_poses, radius, height).astype(np.float32))
            o, d = W.get_rays(poses[-1], intr, H, Wd)
            rays.append((torch.from_numpy(o[lo * Wd:hi * Wd]).to(dev)[None], torch.from_numpy(d[lo * Wd:hi * Wd]).to(dev)[None]))
    else:
        # weak scaling: every rank walks the same orbit, phase-shifted by its rank, so ranks never render the same view at the same step
        for view in sharding.pose_indices(rank, world, n_poses):
            poses.append(W.orbit_pose(view, n_poses * world, radius, height).astype(np.float32))
            o, d = W.get_rays(poses[-1], intr, H, Wd)
            rays.append((torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]))
    N = rays[0][0].shape[1]
    fused = args.path.startswith("fused")
    poses_dev = [torch.from_numpy(p)[None].to(dev) for p in poses]

    def frame(k):
        o, d = rays[k % n_poses]
        if args.path == "fused_camera":                      # secondary: get_rays inside the kernel (ngp_render_frame_camera)
            return ren.render_fused_camera(poses[k % n_poses], intr, H, Wd, dt_gamma=0, bg_color=1, max_steps=1024)
        if args.path == "fused_torch_rays":                  # secondary: what a caller pays with the reference's torch get_rays
            from ngp.nav import get_rays
            r = get_rays(poses_dev[k % n_poses], intr, H, Wd)
            return ren.render_fused(r["rays_o"], r["rays_d"], dt_gamma=0, bg_color=1, max_steps=1024, image_width=Wd)
        if args.path == "fused":
            out = ren.render_fused(o, d, dt_gamma=0, bg_color=1, max_steps=1024, image_width=Wd)
            if strong and args.gather:
                out["full_image"] = sharding.gather_rows(out["image"][0].view(-1, Wd, 3))
            return out
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return ren.run_cuda(o, d, dt_gamma=0, bg_color=1, perturb=False, max_steps=1024,
                                fused_field={"drop_in": None, "per_op": False, "per_op_fused_field": True}[args.path])
            # drop_in: the loop as an unmodified renderer runs it (NGPFieldFF.forward takes its one-launch route by itself); per_op: every op of
            # the field on its own; per_op_fused_field: the one-launch field called explicitly

    def sync_all():
        torch.cuda.synchronize()
        sharding.barrier()                                   # a no-op without a process group (N = 1 and no --force-dist)
        torch.cuda.synchronize()

    stats = []
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # The collector is parked BEFORE the warm-up, not between warm-up and the timed region: gc.collect() + gc.freeze() take ~36 ms of host time during which
    # the GPU idles, and after >= 20 ms of idleness the chip's clocks ramp up again over the next three or four launches (tools/first_launch.py, profiles/HISTORY.md
    # 4.3: first launch +20-28 %, second +8 %, third +4 %; no effect after a gap of 0-1 ms; the workspace's first touch and k_build_coarse are not it).  That
    # ramp was BENCH_r03's `slowest_launch_index: 0` (4.43 against 3.40 ms).
    with no_gc_pauses():
        for k in range(args.warmup):
            frame(k)
        sync_all()
        t0 = time.perf_counter()
        for k in range(args.steps):
            ev[k][0].record()
            out = frame(k)
            ev[k][1].record()
            if fused:
                stats.append(out["stats"])
        sync_all()
        elapsed = time.perf_counter() - t0

    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    if fused:
        samples = int(torch.stack(stats)[:, 0].to(torch.int64).sum().item())
        capped = int(torch.stack(stats)[:, 1].to(torch.int64).sum().item())
    else:
        # op-by-op loop: count with one extra fused pass per pose (same per-ray sample sequence)
        per_pose = [int(ren.render_fused(*rays[k], bg_color=1)["stats"][0].item()) for k in range(n_poses)]
        samples = sum(per_pose[k % n_poses] for k in range(args.steps))
        capped = 0

    samples_all, elapsed_max = sharding.reduce_throughput(samples, elapsed, dev)   # sum of samples, max of wall time

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    value = samples_all / elapsed_max
    ms_per_step = 1e3 * elapsed_max / args.steps
    avg_kernel_s = 1e-3 * float(np.mean(kernel_ms))
    samples_per_launch = samples / args.steps
    achieved = GATHER_BYTES_PER_SAMPLE * samples_per_launch / avg_kernel_s / 1e9
    result = {
        "metric": "ray-samples/sec, Stonehenge-class 800x800 novel-view render",
        "value": value,
        "unit": "ray-samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {
            "workload": f"S-{args.workload} {H}x{Wd} novel-view render (synthetic Stonehenge stand-in, bound 2, dt_gamma 0, max_steps 1024), "
                        f"model={MODEL_NAMES[args.model]}, "
                        f"hashgrid(16x2, 2^19, f16) + FFMLP(32-64-64-16 | 32-64-64-64-16) + occupancy march, path={args.path}",
            "model": args.model,
            "collectives": "nccl (RCCL), forced on a single rank" if (args.force_dist and world == 1) else ("nccl (RCCL)" if use_dist and not rehearsal else ("gloo rehearsal" if use_dist else "none (one process)")),
            "rays_per_frame": N,
            "row_band_per_gpu": f"{hi - lo} of {H} rows" if strong else "all rows",
            "frames_per_gpu": args.steps,
            "samples_per_ray": samples_per_launch / N,
            "poses": n_poses,
        },
        "fps_per_gpu": 1e3 / ms_per_step,
        "rays_capped_at_max_steps": capped,
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "kernel": "k_render_frame_multi" if fused else "per-op loop (many kernels)",
            "avg_launch_ms": 1e3 * avg_kernel_s,
            "launch_ms_min_median_max": [float(np.min(kernel_ms)), float(np.median(kernel_ms)), float(np.max(kernel_ms))],
            "slowest_launch_index": int(np.argmax(kernel_ms)),
            "algorithmic_bytes_per_sample": GATHER_BYTES_PER_SAMPLE,
            "mfma_tflops": MLP_FLOPS_PER_SAMPLE * samples_per_launch / avg_kernel_s / 1e12,
            "mfma_frac": MLP_FLOPS_PER_SAMPLE * samples_per_launch / avg_kernel_s / 1e12 / MFMA_PEAK_TFLOPS,
            # second ceiling: the 25 MB table is Infinity-Cache / L2 resident and read by scattered 4- and 8-byte gathers, so
            # HBM is not what bounds the kernel.  What the texture path processes is lane addresses: 108 per ray-sample (11 hashed levels x 8
            # + 5 dense levels x 4 x-pairs).  The guide documents no peak for that unit; `busy_units` (from the committed PMC profile of these
            # sources) gives the measured busy fractions of TA / TD / VALU instead.
            "gather": {"lane_addresses_per_sample": LANE_ADDRESSES_PER_SAMPLE,
                       "achieved_G_lane_addresses_per_s": LANE_ADDRESSES_PER_SAMPLE * samples_per_launch / avg_kernel_s / 1e9,
                       "l2_line_bytes_per_sample_upper_bound": LINES_PER_SAMPLE * 64},
        },
    }

    # HBM-side traffic per launch: PMC counters cannot be read from inside this process, so the figure comes from the
    # committed rocprofv3 --pmc passes of THIS command (tools/profile_round.sh -> profiles/<tag>_pmc.csv):
    # (FETCH_SIZE + WRITE_SIZE) KiB.  The loads are 4-byte scattered gathers, so the guide's 2x correction for wide
    # coalesced streams does not apply (FETCH_SIZE == TCC_EA0_RDREQ x 64 B in the same profile).  The profile records a hash
    # of the kernel's sources; counters of other sources are refused (traffic stays null).
    if args.path == "fused" and args.res == 800 and args.workload == "ring" and not strong:
        c, info = committed_counters("r[0-9][0-9]_pmc.csv", FRAME_SOURCES)
        if c is not None and "FETCH_SIZE" in c:
            result["roofline"]["traffic"] = (c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024.0
            result["roofline"]["traffic_unit"] = "bytes per launch (L2-miss / fabric side; table is Infinity-Cache resident)"
            result["roofline"]["traffic_source"] = info["profile"]
            if c.get("GRBM_GUI_ACTIVE") and c.get("TA_TA_BUSY_sum"):
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; TA/TD busy over the 256 CUs; SQ_ACTIVE_INST_VALU counts quad-cycles over the
                # 1,024 SIMDs (x4 = cycles): all three as fractions of the cycles their units had during the launch
                cu_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 256.0
                result["roofline"]["busy_units"] = {"source": info["profile"], "ta_busy_frac": c["TA_TA_BUSY_sum"] / cu_cycles,
                                                    "td_busy_frac": c.get("TD_TD_BUSY_sum", 0.0) / cu_cycles,
                                                    "valu_busy_frac": (4.0 * c["SQ_ACTIVE_INST_VALU"] / (4.0 * cu_cycles)) if c.get("SQ_ACTIVE_INST_VALU") else None,
                                                    "mfma_busy_frac": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * cu_cycles)) if c.get("SQ_VALU_MFMA_BUSY_CYCLES") else None,
                                                    "note": "what binds the kernel: the texture path and the VALU, not a memory level"}
            if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") and c.get("TCP_TCC_READ_REQ_sum"):
                # the same launch against the chip's scattered-gather ceilings (tools/micro/gather_rate.hip, profiles/r15_gather_rate.md): the L1 tag pipe
                # (1.55 distinct 64-byte lines per clock and CU), the L2 -> L1 line bandwidth (17 TB/s), the line rate past L2 (4.2 TB/s)
                secs = result["roofline"]["avg_launch_ms"] * 1e-3
                result["roofline"]["memory_levels"] = {
                    "source": info["profile"] + " + profiles/r15_gather_rate.md",
                    "l1_tag_lookups_per_clk_cu": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256.0 * 2.4e9 * secs), "l1_tag_ceiling_per_clk_cu": 1.55,
                    "l2_to_l1_TBps": c["TCP_TCC_READ_REQ_sum"] * 64.0 / secs / 1e12, "l2_to_l1_ceiling_TBps": 17.0,
                    "past_l2_TBps": c["FETCH_SIZE"] * 1024.0 / secs / 1e12, "past_l2_ceiling_TBps": 4.2}
        else:
            result["roofline"]["traffic_stale"] = info

    if args.path == "fused" and world == 1 and not strong and args.frames_per_launch > 1 and H % 8 == 0 and Wd % 8 == 0:
        # secondary: P poses per launch (ngp_render_frames_camera): the same pixels; ramp and drain of the persistent kernel paid once per P frames.
        # The headline `value` stays one frame per launch (BASELINE's metric: a viewer renders the frame it needs now).
        P_ = args.frames_per_launch
        poses_np = np.stack([poses[k % n_poses] for k in range(P_)])
        n_launch = max(args.steps // P_, 3)
        evm, stm = [], []
        with no_gc_pauses():
            for _ in range(2):
                ren.render_fused_cameras(poses_np, intr, H, Wd, dt_gamma=0, bg_color=1, max_steps=1024)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_launch):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                outm = ren.render_fused_cameras(poses_np, intr, H, Wd, dt_gamma=0, bg_color=1, max_steps=1024)
                b.record()
                evm.append((a, b)); stm.append(outm["stats"])
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
        smp = int(torch.stack(stm)[:, 0].to(torch.int64).sum().item())
        k_s = 1e-3 * float(np.mean([a.elapsed_time(b) for a, b in evm]))
        ach = GATHER_BYTES_PER_SAMPLE * (smp / n_launch) / k_s / 1e9
        result["multi_frame"] = {"frames_per_launch": P_, "launches": n_launch, "value": smp / el, "unit": "ray-samples/s", "ms_per_frame": 1e3 * el / (n_launch * P_),
                                 "fps": n_launch * P_ / el, "rays": "generated in the kernel from the poses (camera mode)",
                                 "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                              "kernel": "k_render_frame_multi", "avg_launch_ms": 1e3 * k_s}}
    if fit is not None:
        result["fit"] = fit
    if args.model == "handset" and args.path == "fused" and world == 1 and not strong and not args.no_fit:
        result["fitted"] = fitted_block(args, dev, W, ren, rays, Wd)
        result["train"] = result["fitted"].pop("train")      # config 3's steady state = the last steps of that fit, timed (None with --no-train-block)
    default_line = args.model == "handset" and args.path == "fused" and world == 1 and not strong
    if default_line and not args.no_drop_in_block:
        result["drop_in"] = drop_in_block(args, ren, rays, N)
    if default_line and not args.no_nav_block and args.workload == "ring":
        result["nav"] = nav_block(args, dev, W)
    if not args.no_cpu and world == 1:
        from oracle import ngp_oracle as O, render_oracle as R
        O.build()
        # a one-GPU box grants this process a share of the host (16 cores' worth on the pool this was measured on) whatever the affinity mask says:
        # an OpenMP team of 256 on that share spends its time being descheduled
        cpu_threads = int(os.environ.get("NGP_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
        O.set_threads(cpu_threads)
        if model is None:                                    # the fitted student: hand the oracle its parameters
            f = ren.field
            model = dict(embeddings=f.encoder.embeddings.detach().float().cpu().numpy(), offsets=f.encoder.offsets.cpu().numpy(),
                         per_level_scale=float(f.encoder.per_level_scale), sigma_weights=f.sigma_net.weights.detach().float().cpu().numpy(),
                         color_weights=f.color_net.weights.detach().float().cpu().numpy(), bound=W.BOUND)
        r = args.cpu_res
        o, d = W.get_rays(W.orbit_pose(0, n_poses, radius, height), W.intrinsics(r, r), r, r)
        bitfield = ren.density_bitfield.cpu().numpy()
        t1 = time.perf_counter()
        ref = R.run_cuda(lambda x, dd: R.field_forward(model, x, dd, 1.0), o, d, bitfield, W.BOUND, 2)
        cpu_s = time.perf_counter() - t1
        gpu = ren.render_fused(torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None], bg_color=1)
        img = gpu["image"][0].cpu().numpy()
        result["psnr_vs_oracle_db"] = R.psnr(img, ref["image"])
        result["max_abs_vs_oracle"] = float(np.max(np.abs(img - ref["image"])))
        # BASELINE config 1 ("Lego 64x64, CPU plumbing"): bound 1, one cascade, 64x64 view from the radius-3.2 orbit, timed end to end
        m1 = W.make_model(0, bound=1.0)
        g1 = W.density_grid(bound=1.0)
        bf1, _ = W.bitfield_from_grid(g1)
        o1, d1 = W.get_rays(W.orbit_pose(0, 8, 3.2, 1.2), W.intrinsics(64, 64), 64, 64)
        t2 = time.perf_counter()
        ref1 = R.run_cuda(lambda x, dd: R.field_forward(m1, x, dd, 1.0), o1, d1, bf1, 1.0, 1)
        cfg1_s = time.perf_counter() - t2
        result["cpu_config1"] = {"workload": "64x64 single view, bound 1, oracle run_cuda (BASELINE config 1), end to end", "seconds": cfg1_s,
                                 "ray_samples": ref1["samples"], "value": ref1["samples"] / max(cfg1_s, 1e-9), "unit": "ray-samples/s"}
        result["cpu_baseline"] = {
            "value": ref["samples"] / cpu_s,
            "unit": "ray-samples/s",
            "cores": cpu_threads,
            "kind": "port",
            "sample": f"one {r}x{r} frame of the same scene and model through the oracle's run_cuda, an UNOPTIMISED checker "
                      f"(bit-faithful restatement in chunked numpy + C, the reference has no CPU path): "
                      f"{ref['samples']} ray-samples in {cpu_s:.1f} s, OpenMP team of {cpu_threads}; a stated baseline, not a target",
        }
    print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
