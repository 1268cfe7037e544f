timeout -k 10 600 python -m pytest tests/test_gpu_raymarching.py tests/test_gpu_pipeline.py tests/test_native_modules.py -m gpu -x -q 2>&1 | tail -3
for p in per_op per_op per_op_fused_field; do python bench.py --steps 10 --warmup 3 --no-cpu --path $p 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$p', d['ms_per_step'], d['fps_per_gpu'])"; done
