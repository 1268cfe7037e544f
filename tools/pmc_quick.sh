#!/bin/bash
# (every counter set below has run to completion on this pool; an unknown counter name can hang rocprofv3)
# TCP / SQ counters of the fused frame kernel for one library variant:  bash tools/pmc_quick.sh <variant-name>
set -u
V=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp NGP_HIP_LIB=$R/build/var/libngp_$V.so
cd /tmp
timeout -k 10 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $R/gpurun_out/q_${V}_a -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-fit --no-nav-block --no-drop-in-block --frames-per-launch 1 > $R/gpurun_out/q_${V}_a.log 2>&1
timeout -k 10 150 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum --kernel-trace --output-format csv -d $R/gpurun_out/q_${V}_c -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-fit --no-nav-block --no-drop-in-block --frames-per-launch 1 > $R/gpurun_out/q_${V}_c.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/q_${V}_b -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-fit --no-nav-block --no-drop-in-block --frames-per-launch 1 > $R/gpurun_out/q_${V}_b.log 2>&1
python3 - <<PY
import csv, glob, os, collections
for tag in ("a", "b", "c"):
    fs = sorted(glob.glob("$R/gpurun_out/q_${V}_%s/*/*counter_collection.csv" % tag), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if "k_render_frame" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("$V", k, "%.4g" % (sum(v) / len(v)))
PY
