set -u
O=gpurun_out/r2d; mkdir -p $O
python tools/train_trajectory.py --steps 2400 --every 200 > $O/trajectory.txt 2>&1; cat $O/trajectory.txt | grep step
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r14_perop_kt -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --path per_op > $R/$O/perop_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r14_nav_kt -- python3 $R/tools/time_nav.py --steps 10 --only filter_frozen,planner_frozen > $R/$O/nav_kt.log 2>&1
cd $R; tail -2 $O/perop_kt.log | cut -c1-300; tail -2 $O/nav_kt.log | cut -c1-300
