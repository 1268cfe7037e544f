python -m pytest tests/test_gpu_encoders.py tests/test_gpu_pipeline.py tests/test_gpu_callers_parity.py tests/test_gpu_training_nav.py tests/test_gpu_nav_drivers.py -m gpu -x -q 2>&1 | tail -3
for p in per_op per_op_fused_field; do python bench.py --steps 10 --warmup 3 --no-cpu --path $p 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$p', d['ms_per_step'], d['fps_per_gpu'])"; done
python bench.py --mode train --steps 32 --warmup 8 --settle 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train warm ms/step', d['warmup_phase']['ms_per_step'], 'second phase', d['ms_per_step'])"
python tools/time_grid_fwd.py 2>&1 | tail -4
