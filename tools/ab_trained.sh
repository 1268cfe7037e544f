# A/B of render_fused.hip build variants on the TRAINED model: bash tools/ab_trained.sh v1 v2 ...   (FIT=steps, default 8000)
for v in "$@"; do
  NGP_HIP_LIB=$GRAFT_REPO_ROOT/build/var/libngp_$v.so timeout -k 10 300 python bench.py --model trained --fit-steps ${FIT:-8000} --steps ${STEPS:-50} --warmup ${WARMUP:-10} --no-cpu 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$v', round(r['ms_per_step'],2), 'ms', round(r['config']['samples_per_ray'],1), 'samples/ray', round(r['value']/1e6), 'Msamples/s', 'frac', round(r['roofline']['frac'],3))"
done
