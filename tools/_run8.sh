export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r14_perop2_kt -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --path per_op > $R/gpurun_out/r14_perop2.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/r14_perop2_kt/*/*kernel_stats.csv'))[-1]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms per frame', tot/1e6/8)
for r in rows[:14]: print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us', round(float(r['TotalDurationNs'])/1e6/8,2),'ms/frame')
f=sorted(glob.glob('gpurun_out/r14_perop2_kt/*/*kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f)) if 'k_march_rays' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print([round(x) for x in d[-64:]])
PY
