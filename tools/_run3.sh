set -u
O=gpurun_out/r2e; mkdir -p $O
python -m pytest tests/test_gpu_raymarching.py tests/test_gpu_pipeline.py tests/test_gpu_ffmlp.py tests/test_gpu_encoders.py tests/test_native_modules.py -m gpu -x -q 2>&1 | tail -4
python tools/time_field.py 2>&1 | grep "M ="
python bench.py --steps 10 --warmup 3 --no-cpu --path per_op 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('per_op ms/frame', d['ms_per_step'], 'fps', d['fps_per_gpu'])"
python bench.py --steps 10 --warmup 3 --no-cpu --path per_op_fused_field 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('per_op_fused_field ms/frame', d['ms_per_step'], 'fps', d['fps_per_gpu'])"
