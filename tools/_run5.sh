export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r14_navn_kt -- python3 $R/tools/time_nav.py --steps 10 --only filter_native,planner_native > $R/gpurun_out/r14_navn.log 2>&1
cd $R; tail -2 gpurun_out/r14_navn.log | cut -c1-200
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/r14_navn_kt/*/*kernel_stats.csv'))[-1]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:8]: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
