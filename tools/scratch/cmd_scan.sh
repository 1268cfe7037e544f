set -e
timeout -k 10 300 python -m pytest tests/test_gpu_field_train.py tests/test_gpu_adam.py tests/test_gpu_callers_golden.py tests/test_gpu_grid_scatter.py tests/test_gpu_church.py -x -q -m gpu > gpurun_out/scan_tests.log 2>&1 || { tail -30 gpurun_out/scan_tests.log; exit 1; }
tail -2 gpurun_out/scan_tests.log
for i in 1 2; do
timeout -k 10 200 python bench.py --mode train --steps 50 --warmup 20 --settle 1000 > gpurun_out/scan_train_$i.json 2> gpurun_out/scan_train_$i.err
python - <<PY
import json
r=json.loads(open("gpurun_out/scan_train_$i.json").read().strip().splitlines()[-1])
print("steady", r["ms_per_step"], "early", r["warmup_phase"]["ms_per_step"], "loss", r["config"]["final_loss"], r["warmup_phase"]["loss"])
PY
done
