#!/bin/bash
# Counters of the field's training launches (forward / backward) on the early-phase batch of tools/time_field_train.py.
# FETCH_SIZE and WRITE_SIZE each need a pass of their own (together they hang rocprofv3 on this pool).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
SETS=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_MFMA"
)
for ((i=0; i<${#SETS[@]}; i++)); do
  rm -rf $R/gpurun_out/ft_pmc$i
  timeout -k 10 150 rocprofv3 --pmc ${SETS[$i]} --kernel-trace --output-format csv -d $R/gpurun_out/ft_pmc$i -- python3 $R/tools/time_field_train.py --iters 2 > $R/gpurun_out/ft_pmc$i.log 2>&1
  echo "pmc pass $i done rc=$?"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/ft_pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for name in ("k_ft_encode_levels", "k_field_train_forward", "k_field_train_backward<0", "k_field_train_backward<1", "k_gs_bin", "k_gs_accumulate", "k_grid_forward"):
            if name in k:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    for c, v in sorted(cs.items()):
        print(name, c, "%.5g" % (sum(v) / len(v)), "n=%d" % len(v))
PY
