set -u
bash tools/profile_round.sh r14 full -- bench.py --steps 30 --warmup 5 --no-cpu 2>&1 | tail -2
bash tools/profile_round.sh r14_train short -- bench.py --mode train --steps 16 --warmup 4 --settle 64 2>&1 | tail -1
bash tools/profile_round.sh r14_perop short -- bench.py --steps 6 --warmup 2 --no-cpu --path per_op 2>&1 | tail -1
bash tools/profile_round.sh r14_nav short -- tools/time_nav.py --only filter_native,filter_frozen,planner_frozen 2>&1 | tail -1
ls gpurun_out | grep r14 | head -40
