set -e
timeout -k 10 400 python -m pytest tests/test_gpu_field_train.py tests/test_gpu_grid_scatter.py tests/test_gpu_adam.py tests/test_gpu_callers_golden.py tests/test_gpu_church.py tests/test_gpu_bench_rehearsal.py -x -q -m gpu > gpurun_out/ft_tests.log 2>&1 || { tail -40 gpurun_out/ft_tests.log; exit 1; }
tail -2 gpurun_out/ft_tests.log
for cfg in 1 0 1; do
NGP_FT_LIVE_ONLY=$cfg timeout -k 10 200 python bench.py --mode train --steps 50 --warmup 20 --settle 1000 > gpurun_out/ab_$cfg.json 2> gpurun_out/ab_$cfg.err
python - <<PY
import json
r=json.loads(open("gpurun_out/ab_$cfg.json").read().strip().splitlines()[-1])
print("live $cfg: steady", r["ms_per_step"], "early", r["warmup_phase"]["ms_per_step"], "loss", r["config"]["final_loss"], r["warmup_phase"]["loss"], "points", r["config"]["points_per_step"], "scatter", r["roofline"]["avg_launch_ms"], r["warmup_phase"]["roofline"]["avg_launch_ms"])
PY
done
