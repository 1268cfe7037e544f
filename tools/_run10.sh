set -u
O=gpurun_out/r2f; mkdir -p $O
python bench.py > $O/bench_render.json 2> $O/err.txt; python - <<'PY'
import json; d=json.load(open('gpurun_out/r2f/bench_render.json')); r=d['roofline']
print('render', d['value']/1e9, 'G samples/s', d['ms_per_step'], 'ms', 'frac', r['frac'], 'traffic', r.get('traffic'), 'busy', r.get('busy_units'), 'stale', r.get('traffic_stale'), 'gather', r['gather'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'psnr', d['psnr_vs_oracle_db'])
PY
python bench.py --steps 30 --warmup 5 --no-cpu --workload church > $O/bench_church.json 2>> $O/err.txt; python -c "
import json; d=json.load(open('gpurun_out/r2f/bench_church.json')); print('church', d['value']/1e9, d['ms_per_step'], d['config']['samples_per_ray'], d['roofline']['frac'])"
python bench.py --mode train --steps 64 --warmup 8 --settle 1500 > $O/bench_train.json 2>> $O/err.txt; python -c "
import json; d=json.load(open('gpurun_out/r2f/bench_train.json')); print('train steady', d['ms_per_step'], d['value'], d['config']['points_per_step'], d['roofline']['frac'], d['roofline'].get('profiled_requests_per_launch'), 'warm', d['warmup_phase']['ms_per_step'], d['warmup_phase']['points_per_step'], d['warmup_phase']['roofline']['frac'])"
python bench.py --mode train --steps 32 --warmup 8 --settle 600 --workload church > $O/bench_train_church.json 2>> $O/err.txt; python -c "
import json; d=json.load(open('gpurun_out/r2f/bench_train_church.json')); print('train church steady', d['ms_per_step'], d['config']['points_per_step'], 'warm', d['warmup_phase']['ms_per_step'])"
for p in per_op per_op_fused_field fused_camera; do python bench.py --steps 10 --warmup 3 --no-cpu --path $p 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$p', d['ms_per_step'], d['fps_per_gpu'])"; done
python tools/time_nav.py --steps 30 2>&1 | tail -1
python tools/time_grid_update.py 2>&1 | grep update
