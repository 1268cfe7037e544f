"""Per-kernel GPU time per step over the LAST N dispatches of each kernel of a rocprofv3 --kernel-trace CSV (a run whose early steps differ from its
steady state): python tools/last_steps_kernel_stats.py <kernel_trace.csv> <steps> [top]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ref = [r for r in rows if "k_composite_train_fwd" in r["Kernel_Name"]]
t0 = int(ref[-steps]["Start_Timestamp"])
acc = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) >= t0:
        a = acc[r["Kernel_Name"][:90]]
        a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
tot = sum(v[0] for v in acc.values())
print(f"last {steps} steps: {tot / steps / 1e3:.1f} us of GPU time per step, wall {(int(rows[-1]['End_Timestamp']) - t0) / steps / 1e3:.1f} us per step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{v[0] / steps / 1e3:8.1f} us/step  {v[1] / steps:5.1f} launches/step  {k}")
