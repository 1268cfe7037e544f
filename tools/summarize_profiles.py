#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs that tools/profile_round.sh left in gpurun_out/ into the small files kept under profiles/.

    python tools/summarize_profiles.py r01

writes profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 --stats table), profiles/<tag>_pmc.csv (per-dispatch mean of
every counter for the render kernel) and profiles/<tag>_summary.md (derived figures, formulas stated)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

def newest(pattern):
    """rocprofv3 names its files by PID and gpurun merges into an existing directory: keep the most recent run only"""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


ks = newest(os.path.join(src, f"{tag}_kt", "*", "*kernel_stats.csv"))
assert ks, "no kernel-trace stats found"
shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
stats = {r["Name"]: r for r in csv.DictReader(open(ks[0]))}
kname = next(n for n in stats if "k_render_frame" in n)
avg_ms = float(stats[kname]["AverageNs"]) / 1e6

counters = {}
for f in sorted(sum((newest(os.path.join(d, "*", "*counter_collection.csv")) for d in glob.glob(os.path.join(src, f"{tag}_pmc*"))), [])):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_render_frame" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        counters[k] = sum(v.values()) / len(v)
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w") as f:
    f.write("counter,mean_per_dispatch\n")
    for k in sorted(counters):
        f.write(f"{k},{counters[k]:.6g}\n")

bench = None
bj = os.path.join(src, f"{tag}_bench_under_profiler.json")
if os.path.exists(bj) and os.path.getsize(bj):
    bench = json.loads(open(bj).read())
c = counters.get
lines = [f"# {tag}: rocprofv3 summary for `k_render_frame_multi` (bench.py, 800x800 S-ring, 1x MI355X)", ""]
lines.append(f"* kernel-trace --stats: **{avg_ms:.3f} ms** average over {stats[kname]['Calls']} launches "
             f"({float(stats[kname]['Percentage']):.2f} % of GPU time)")
if bench:
    lines.append(f"* bench.py under the profiler: {bench['ms_per_step']:.3f} ms/step, event-timed launch {bench['roofline']['avg_launch_ms']:.3f} ms, "
                 f"{bench['config']['samples_per_ray'] * bench['config']['rays_per_frame'] / 1e6:.2f} M ray-samples per launch")
if c("FETCH_SIZE") is not None:
    fetch, write = c("FETCH_SIZE") * 1024, (c("WRITE_SIZE") or 0) * 1024
    lines.append(f"* fabric traffic per launch: FETCH_SIZE {fetch / 1e9:.2f} GB + WRITE_SIZE {write / 1e9:.3f} GB "
                 f"(= TCC_EA0_RDREQ x 64 B: {c('TCC_EA0_RDREQ_sum', 0) * 64 / 1e9:.2f} GB; 4-byte scattered loads, so the guide's 2x "
                 f"wide-stream correction does not apply; the 25 MB table lives in the Infinity Cache, so this is L2-miss traffic, an upper bound on HBM bytes)")
if c("TCC_REQ_sum"):
    lines.append(f"* L2: {c('TCC_REQ_sum') / 1e9:.2f} G requests, hit rate {c('TCC_HIT_sum') / (c('TCC_HIT_sum') + c('TCC_MISS_sum')):.3f}")
if c("TCP_TOTAL_CACHE_ACCESSES_sum"):
    lines.append(f"* L1 (TCP): {c('TCP_TOTAL_ACCESSES_sum') / 1e9:.2f} G accesses, {c('TCP_TCC_READ_REQ_sum') / 1e9:.2f} G read requests to L2, "
                 f"pending-stall cycles {c('TCP_PENDING_STALL_CYCLES_sum') / 1e9:.2f} G summed over 256 CUs")
if c("SQ_WAVE_CYCLES"):
    lines.append(f"* SQ: wait-any / wave-cycles = {c('SQ_WAIT_ANY') / c('SQ_WAVE_CYCLES'):.3f}, VALU-active / wave-cycles = "
                 f"{c('SQ_ACTIVE_INST_VALU') / c('SQ_WAVE_CYCLES'):.3f}, {c('SQ_INSTS_VALU') / 1e9:.2f} G VALU and "
                 f"{c('SQ_INSTS_VMEM_RD') / 1e6:.1f} M vector-memory-read wave-instructions per launch")
if c("SQ_INSTS_MFMA"):
    lines.append(f"* MFMA: {c('SQ_INSTS_MFMA') / 1e6:.1f} M instructions, busy cycles {c('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1e9:.2f} G; "
                 f"LDS: {c('SQ_INSTS_LDS', 0) / 1e6:.1f} M instructions, bank-conflict cycles {c('SQ_LDS_BANK_CONFLICT', 0) / 1e6:.1f} M")
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
