"""Launch-by-launch account of ONE frame of the unchanged-caller loop (`bench.py --path drop_in`: nerf/renderer.py:325-374 over the drop-in ops) from a
rocprofv3 --kernel-trace CSV: per kernel name launches / GPU time, the GPU-idle gaps between consecutive launches split by size, and the first iterations
launch by launch.  Frames are delimited by k_near_far launches (one per run_cuda call).
   python tools/drop_in_timeline.py <kernel_trace.csv> [frame-from-the-end, default 1] [iterations to list, default 2]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_list = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_near_far" in r["Kernel_Name"]]
a = marks[-back - 1]
b = marks[-back]
frame = rows[a:b]
t0 = int(frame[0]["Start_Timestamp"])
t1 = int(rows[b]["Start_Timestamp"])
per = collections.defaultdict(lambda: [0, 0.0])
gaps, prev = [], None
iters = 0
for r in frame:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:70]
    per[name][0] += 1
    per[name][1] += (e - s) / 1e3
    if prev is not None:
        gaps.append((max(s - prev, 0) / 1e3, name))
    prev = e
    if "k_march_rays" in name and "train" not in name:
        iters += 1
busy = sum(v[1] for v in per.values())
print(f"frame: {(t1 - t0) / 1e3:.1f} us launch to launch, {len(frame)} launches, {iters} loop iterations, GPU busy {busy:.1f} us, idle {(t1 - t0) / 1e3 - busy:.1f} us")
print("\nper kernel (launches, total us, mean us):")
for name, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:5d} {us:9.1f} {us / n:7.2f}  {name}")
print("\nidle gaps before a launch, by size:")
for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)):
    sel = [g for g, _ in gaps if lo <= g < hi]
    print(f"  {lo:>3} .. {hi if hi < 1e9 else 'inf':>4} us: {len(sel):5d} gaps, {sum(sel):9.1f} us")
big = collections.defaultdict(lambda: [0, 0.0])
for g, name in gaps:
    if g >= 10:
        big[name][0] += 1
        big[name][1] += g
print("\ngaps >= 10 us, by the kernel that follows them:")
for name, (n, us) in sorted(big.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {n:5d} {us:9.1f}  {name}")
print(f"\nfirst {n_list} iterations, launch by launch (start us, duration, gap before):")
seen, prev = 0, None
for r in frame:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:90]
    if "k_march_rays" in name and "train" not in name:
        seen += 1
        if seen > n_list:
            break
    print("  %9.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, name))
    prev = e
