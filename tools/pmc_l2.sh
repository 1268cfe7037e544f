#!/bin/bash
# L2 / fabric counters of the fused frame kernel for library variants:  bash tools/pmc_l2.sh <variant> [<variant> ...]
# (the counter set is one of tools/profile_round.sh's, which has run to completion on this pool)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for V in "$@"; do
  export NGP_HIP_LIB=$R/build/var/libngp_$V.so
  timeout -k 10 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/l2_${V} -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-fit --frames-per-launch 1 > $R/gpurun_out/l2_${V}.log 2>&1
  python3 - <<PY
import csv, glob, os, collections
fs = sorted(glob.glob("$R/gpurun_out/l2_${V}/*/*counter_collection.csv"), key=os.path.getmtime)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[-1])):
    if "k_render_frame" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$V", {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
done
