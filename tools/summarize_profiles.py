#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs that tools/profile_round.sh left in gpurun_out/ into the small files kept under profiles/.

    python tools/summarize_profiles.py <tag> [kernel-substring ...] [--title "what ran"] [--top N]

writes profiles/<tag>_kernel_stats.csv (the rocprofv3 --stats table; --top N keeps the N longest kernels), profiles/<tag>_pmc.csv
(per-dispatch mean of every counter for the FIRST named kernel: the file bench.py reads `roofline.traffic` from),
profiles/<tag>_meta.json (hash of the kernel sources of this tree: bench.py refuses the counters of other sources) and
profiles/<tag>_summary.md (derived figures, formulas stated).  Refuses to run when the tree's sources differ from the library that
was profiled (gpurun_out/<tag>_sources.txt, written on the GPU box) -- a stale summary cannot be produced by accident."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("kernels", nargs="*", default=["k_render_frame"])
ap.add_argument("--title", default="bench.py, 800x800 S-ring, 1x MI355X")
ap.add_argument("--top", type=int, default=0)
ap.add_argument("--dst", default=None, help="output directory (default: profiles/ of the repository)")
ap.add_argument("--last", type=int, default=0, help="counter means over the LAST N dispatches of the first kernel only (a run that launches it on other inputs first)")
ap.add_argument("--sources", default="frame", choices=["frame", "train"], help="which kernel's source set the recorded hash covers")
args = ap.parse_args()
tag = args.tag
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), (args.dst or os.path.join(root, "profiles"))
os.makedirs(dst, exist_ok=True)


sys.path.insert(0, root)
import bench  # noqa: E402  (sources_sha16 and the per-kernel source lists live there)

SOURCE_SETS = {"frame": bench.FRAME_SOURCES, "train": bench.TRAIN_SOURCES}


def sources_sha16():
    return bench.sources_sha16(SOURCE_SETS[args.sources])


def newest(pattern):
    """rocprofv3 names its files by PID and gpurun merges into an existing directory: keep the most recent run only"""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


sha = sources_sha16()
stamp = os.path.join(src, f"{tag}_sources.txt")
if os.path.exists(stamp):
    profiled = open(stamp).read().split()[0 if args.sources == "frame" else 1]
    if profiled != sha:
        sys.exit(f"refusing: {tag} was profiled with kernel sources {profiled}, this tree has {sha} (re-profile, or check out the profiled commit)")

ks = newest(os.path.join(src, f"{tag}_kt", "*", "*kernel_stats.csv"))
assert ks, "no kernel-trace stats found"
rows = list(csv.DictReader(open(ks[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
keep = rows[:args.top] if args.top else rows
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(keep)
stats = {r["Name"]: r for r in rows}
total_ns = sum(float(r["TotalDurationNs"]) for r in rows)


def find(sub):
    return next((n for n in stats if sub in n), None)


counters = {}
first = find(args.kernels[0])
for f in sorted(sum((newest(os.path.join(d, "*", "*counter_collection.csv")) for d in glob.glob(os.path.join(src, f"{tag}_pmc*"))), [])):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if args.kernels[0] in r["Kernel_Name"]:
            agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        vals = [v[d] for d in sorted(v, key=int)]
        if args.last:
            vals = vals[-args.last:]
        counters[k] = sum(vals) / len(vals)
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w") as f:
    f.write("counter,mean_per_dispatch\n")
    for k in sorted(counters):
        f.write(f"{k},{counters[k]:.6g}\n")
json.dump({"tag": tag, "sources_sha16": sha, "kernel": first, "title": args.title}, open(os.path.join(dst, f"{tag}_meta.json"), "w"))

bench = None
bj = os.path.join(src, f"{tag}_bench_under_profiler.json")
if os.path.exists(bj) and os.path.getsize(bj):
    try:
        bench = json.loads(open(bj).read())
    except ValueError:
        bench = None
c = counters.get
lines = [f"# {tag}: rocprofv3 summary ({args.title}); kernel sources {sha}", ""]
lines.append(f"* all kernels: {total_ns / 1e6:.2f} ms of GPU time in {sum(int(r['Calls']) for r in rows)} launches")
for sub in args.kernels:
    n = find(sub)
    if n is None:
        lines.append(f"* `{sub}`: not launched")
        continue
    r = stats[n]
    lines.append(f"* `{sub}` kernel-trace --stats: **{float(r['AverageNs']) / 1e6:.4f} ms** average over {r['Calls']} launches "
                 f"({float(r['Percentage']):.2f} % of GPU time, total {float(r['TotalDurationNs']) / 1e6:.2f} ms)")
if bench and "ms_per_step" in bench:
    extra = ""
    if bench.get("roofline") and bench["roofline"].get("avg_launch_ms"):
        extra = f", event-timed launch of {bench['roofline'].get('kernel', 'the kernel')} {bench['roofline']['avg_launch_ms']:.3f} ms, roofline frac {bench['roofline']['frac']:.3f}"
    lines.append(f"* the command's own line under the profiler: {bench['ms_per_step']:.3f} ms/step{extra}")
if c("FETCH_SIZE") is not None:
    fetch, write = c("FETCH_SIZE") * 1024, (c("WRITE_SIZE") or 0) * 1024
    if args.last:
        lines.append(f"* counters below: means over the last {args.last} dispatches of `{args.kernels[0]}`")
    lines.append(f"* `{args.kernels[0]}` fabric traffic per launch: FETCH_SIZE {fetch / 1e9:.3f} GB + WRITE_SIZE {write / 1e9:.3f} GB "
                 f"(TCC_EA0_RDREQ x 64 B = {(c('TCC_EA0_RDREQ_sum') or 0) * 64 / 1e9:.3f} GB; scattered 4/8-byte accesses, so the guide's 2x "
                 f"wide-stream correction does not apply; L2-miss traffic, an upper bound on HBM bytes)")
if c("TCC_REQ_sum"):
    lines.append(f"* L2: {c('TCC_REQ_sum') / 1e9:.3f} G requests, hit rate {c('TCC_HIT_sum') / max(c('TCC_HIT_sum') + c('TCC_MISS_sum'), 1):.3f}")
if c("TCP_TOTAL_CACHE_ACCESSES_sum"):
    lines.append(f"* L1 (TCP): {c('TCP_TOTAL_ACCESSES_sum') / 1e9:.3f} G accesses, {c('TCP_TCC_READ_REQ_sum') / 1e9:.3f} G read requests to L2, "
                 f"pending-stall cycles {c('TCP_PENDING_STALL_CYCLES_sum') / 1e9:.3f} G summed over the CUs")
if c("SQ_WAVE_CYCLES"):
    lines.append(f"* SQ: wait-any / wave-cycles = {c('SQ_WAIT_ANY') / c('SQ_WAVE_CYCLES'):.3f}, VALU-active / wave-cycles = "
                 f"{c('SQ_ACTIVE_INST_VALU') / c('SQ_WAVE_CYCLES'):.3f}, {c('SQ_INSTS_VALU') / 1e9:.3f} G VALU and "
                 f"{c('SQ_INSTS_VMEM_RD') / 1e6:.2f} M vector-memory-read wave-instructions per launch")
if c("SQ_INSTS_MFMA") is not None:
    lines.append(f"* MFMA: {c('SQ_INSTS_MFMA') / 1e6:.2f} M instructions, busy cycles {(c('SQ_VALU_MFMA_BUSY_CYCLES') or 0) / 1e9:.3f} G; "
                 f"LDS: {(c('SQ_INSTS_LDS') or 0) / 1e6:.2f} M instructions, bank-conflict cycles {(c('SQ_LDS_BANK_CONFLICT') or 0) / 1e6:.2f} M")
if c("TA_TA_BUSY_sum"):
    lines.append(f"* texture path: TA busy {c('TA_TA_BUSY_sum') / 1e9:.3f} G, TD busy {(c('TD_TD_BUSY_sum') or 0) / 1e9:.3f} G cycles summed over the CUs "
                 f"(GRBM_GUI_ACTIVE {(c('GRBM_GUI_ACTIVE') or 0) / 1e6:.2f} M cycles)")
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
