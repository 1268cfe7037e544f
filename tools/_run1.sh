set -u
O=gpurun_out/r2c; mkdir -p $O
python bench.py --steps 50 --warmup 10 > $O/bench_render.json 2> $O/bench_render.err; tail -c 2500 $O/bench_render.json; echo
python bench.py --steps 30 --warmup 5 --no-cpu --workload church > $O/bench_church.json 2>> $O/bench_render.err; tail -c 1200 $O/bench_church.json; echo
python bench.py --mode train --steps 32 --warmup 8 --settle 160 > $O/bench_train.json 2> $O/bench_train.err; cat $O/bench_train.json
python bench.py --steps 10 --warmup 3 --no-cpu --path per_op > $O/bench_perop.json 2>&1; tail -c 600 $O/bench_perop.json; echo
python tools/time_nav.py --steps 20 > $O/time_nav.txt 2>&1; cat $O/time_nav.txt
bash tools/profile_round.sh r14_train short -- bench.py --mode train --steps 16 --warmup 4 --settle 48 > $O/prof_train.log 2>&1; tail -3 $O/prof_train.log
