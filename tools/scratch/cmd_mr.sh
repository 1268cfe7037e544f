set -e
timeout -k 10 400 python -m pytest tests/test_gpu_raymarching.py tests/test_gpu_callers_golden.py tests/test_gpu_callers_parity.py tests/test_gpu_nav_golden.py -x -q -m gpu > gpurun_out/mr_tests.log 2>&1 || { tail -40 gpurun_out/mr_tests.log; exit 1; }
tail -2 gpurun_out/mr_tests.log
timeout -k 10 200 python tools/march_schedule.py > gpurun_out/march_schedule_new.log 2>&1 || true
tail -12 gpurun_out/march_schedule_new.log
timeout -k 10 200 python bench.py --path drop_in --steps 10 --warmup 3 --no-cpu --no-fit --no-nav-block --no-drop-in-block --frames-per-launch 1 > gpurun_out/dropin_new.json 2> gpurun_out/dropin_new.err
python - <<PY
import json
r=json.loads(open("gpurun_out/dropin_new.json").read().strip().splitlines()[-1])
print("drop_in ms per frame", r["ms_per_step"])
PY
