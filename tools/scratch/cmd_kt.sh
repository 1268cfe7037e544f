set -e
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_tmp -- python3 $R/bench.py --mode train --steps 16 --warmup 4 --settle 64 --no-cpu > $R/gpurun_out/kt_tmp.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt_tmp/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    print("%-70s %6s %10.1f us avg  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf gpurun_out/kt_tmp
