#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box:   gpurun -- 'bash tools/profile_round.sh r01'
# 1) --kernel-trace --stats            -> per-kernel durations (must agree with bench.py's event timing)
# 2) separate --pmc passes (never combined with other trace domains): SQ mix, L2 (TCC), FETCH_SIZE, WRITE_SIZE, L1 (TCP), MFMA/LDS
# Raw CSVs land in gpurun_out/<tag>_*; tools/summarize_profiles.py turns them into profiles/<tag>_*.{csv,md}.
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu > $R/gpurun_out/${TAG}_kt.log 2>&1
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_MFMA" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_IFETCH SQ_IFETCH_LEVEL" \
  "GRBM_GUI_ACTIVE GRBM_COUNT TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu > $R/gpurun_out/${TAG}_pmc$i.log 2>&1
done
grep -h '^{' $R/gpurun_out/${TAG}_kt.log | tail -1 > $R/gpurun_out/${TAG}_bench_under_profiler.json
echo "profiles collected under gpurun_out/${TAG}_*"
