#!/bin/bash
# Collect the rocprofv3 evidence for one command on the GPU box:
#   gpurun -- 'bash tools/profile_round.sh <tag> [short|full] -- bench.py --steps 30 --warmup 5 --no-cpu'
# 1) --kernel-trace --stats            -> per-kernel durations (must agree with the command's own event timing)
# 2) separate --pmc passes (never combined with other trace domains): SQ mix, L2 (TCC), FETCH_SIZE, WRITE_SIZE, L1 (TCP), MFMA/LDS, TA/TD
#    ("short" = the four passes the summary's traffic / L2 / instruction-mix lines need; "full" = all eight)
# The program after `--` is run as `python3 <args>` directly under rocprofv3 (no env / bash hop: the profiler initialises the GPU first).
# Raw CSVs land in gpurun_out/<tag>_*; tools/summarize_profiles.py <tag> <kernel-substring> condenses them into profiles/<tag>_*.
set -u
TAG=${1:-r01}; shift
MODE=full
if [ "${1:-}" = "short" ] || [ "${1:-}" = "full" ]; then MODE=$1; shift; fi
if [ "${1:-}" = "--" ]; then shift; fi
if [ $# -eq 0 ]; then set -- bench.py --steps 30 --warmup 5 --no-cpu; fi
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
SCRIPT=$R/$1; shift
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.sources_sha16(bench.FRAME_SOURCES), bench.sources_sha16(bench.TRAIN_SOURCES))" > $R/gpurun_out/${TAG}_sources.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -- python3 $SCRIPT "$@" > $R/gpurun_out/${TAG}_kt.log 2>&1
echo "kernel trace done: $(grep -c . $R/gpurun_out/${TAG}_kt.log) log lines"
SETS=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_MFMA"
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_IFETCH SQ_IFETCH_LEVEL"
  "GRBM_GUI_ACTIVE GRBM_COUNT TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
)
N=${#SETS[@]}
if [ "$MODE" = "short" ]; then N=4; fi
# PMC passes replay every dispatch: keep them short (the caller's --steps / --warmup are overridden when the script takes them)
PMCARGS=()
for a in "$@"; do PMCARGS+=("$a"); done
for ((i=0; i<N; i++)); do
  rocprofv3 --pmc ${SETS[$i]} --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc$((i+1)) -- python3 $SCRIPT "${PMCARGS[@]}" --steps 4 --warmup 1 > $R/gpurun_out/${TAG}_pmc$((i+1)).log 2>&1
  echo "pmc pass $((i+1))/$N done"
done
grep -h '^{' $R/gpurun_out/${TAG}_kt.log | tail -1 > $R/gpurun_out/${TAG}_bench_under_profiler.json
echo "profiles collected under gpurun_out/${TAG}_*"
