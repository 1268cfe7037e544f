python -m pytest tests/test_gpu_nav_native.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2>gpurun_out/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print({k: d[k] for k in ('value','ms_per_step','cpu_config1')}, d['cpu_baseline'], d['roofline']['busy_units'])"
python bench.py --steps 20 --warmup 5 --no-cpu --model trained --fit-steps 2000 > gpurun_out/bench_trained.json 2>>gpurun_out/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/bench_trained.json')); print('trained', d['value']/1e9, d['ms_per_step'], d['config']['samples_per_ray'], d['fit'], d['roofline']['frac'])"
