python -m pytest tests/test_gpu_fused_variants.py -m gpu -x -q 2>&1 | tail -3
for p in fused fused_camera fused fused_camera; do python bench.py --steps 60 --warmup 10 --no-cpu --path $p 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$p', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['avg_launch_ms'],3))"; done
