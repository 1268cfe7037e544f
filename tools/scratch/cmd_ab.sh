set -e
for cfg in "1 1" "0 1" "1 0" "0 0"; do
set -- $cfg
NGP_FT_LIVE_ONLY=$1 NGP_GS_SKIP_DEAD=$2 timeout -k 10 200 python bench.py --mode train --steps 50 --warmup 20 --settle 1000 > gpurun_out/ab_$1$2.json 2> gpurun_out/ab_$1$2.err
python - <<PY
import json
r=json.loads(open("gpurun_out/ab_$1$2.json").read().strip().splitlines()[-1])
print("live $1 skip $2: steady", r["ms_per_step"], "early", r["warmup_phase"]["ms_per_step"], "loss", r["config"]["final_loss"], r["warmup_phase"]["loss"], "points", r["config"]["points_per_step"])
PY
done
