# A/B of render_fused.hip build variants in one gpurun call: bash tools/ab_variants.sh v1 v2 ...
# prints ms per frame, ray-samples per frame and Msamples/s (timing-only builds change the sample count)
for v in "$@"; do
  NGP_HIP_LIB=$GRAFT_REPO_ROOT/build/var/libngp_$v.so timeout -k 10 120 python bench.py --steps ${STEPS:-16} --warmup ${WARMUP:-3} --no-cpu 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$v', round(r['ms_per_step'],2), 'ms', round(r['config']['samples_per_ray']*r['config']['rays_per_frame']/1e6,2), 'Msamples/frame', round(r['value']/1e6), 'Msamples/s')"
done
