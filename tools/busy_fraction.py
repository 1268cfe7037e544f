"""GPU-busy fraction of the tail of a rocprofv3 kernel trace: sum of kernel durations / wall time over the last `--ms` milliseconds of the run.
   python tools/busy_fraction.py <kernel_trace.csv> [--ms 150]"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--ms", type=float, default=150.0)
ap.add_argument("--skip-ms", type=float, default=30.0, help="ignore this much at the very end (teardown, the bench's own reductions)")
args = ap.parse_args()
rows = list(csv.DictReader(open(args.trace)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
end = ev[-1][1] - int(args.skip_ms * 1e6)
lo = end - int(args.ms * 1e6)
win = [(a, b, n) for a, b, n in ev if a >= lo and b <= end]
busy = sum(b - a for a, b, _ in win)
print(f"window {args.ms:.0f} ms: {len(win)} launches, GPU busy {busy / 1e6:.1f} ms = {busy / (args.ms * 1e6):.2f} of wall")
by = {}
for a, b, n in win:
    k = n.split("(")[0][:70]
    by[k] = by.get(k, 0) + (b - a)
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v / 1e6:7.2f} ms  {k}")
