"""One training step of a rocprofv3 --kernel-trace CSV, launch by launch: start (us since the step's compositor forward), duration, gap to the previous
launch's end.  python tools/one_step_timeline.py <kernel_trace.csv> [steps-from-the-end, default 3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ref = [i for i, r in enumerate(rows) if "k_composite_train_fwd" in r["Kernel_Name"]]
a, b = ref[-back], ref[-back + 1]
t0, prev = int(rows[a]["Start_Timestamp"]), None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r["Kernel_Name"][:100]))
    prev = e
print("step: %.1f us from launch to launch" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
