# Collect and summarise the tracked profiles on the GPU box in one gpurun call (TAG=rNN, default r16); summaries land in gpurun_out/profiles_$TAG (copy them to profiles/).
#   gpurun -- 'bash tools/profile_all.sh [frame] [train] [steady] [perop] [nav] [trained]'       (default: all but steady)
set -e
TAG=${TAG:-r16}
D=gpurun_out/profiles_${TAG}
WHAT="${@:-frame train perop nav trained}"
for w in $WHAT; do
  case $w in
    frame)
      bash tools/profile_round.sh ${TAG} full -- bench.py --steps 30 --warmup 5 --no-cpu --no-fit --no-nav-block --no-drop-in-block --frames-per-launch 1
      python tools/summarize_profiles.py ${TAG} k_render_frame --dst $D --top 12 > gpurun_out/sum_${TAG}.log 2>&1
      rm -rf gpurun_out/${TAG}_kt gpurun_out/${TAG}_pmc? ;;
    train)
      bash tools/profile_round.sh ${TAG}_train short -- bench.py --mode train --steps 16 --warmup 4 --settle 64 --no-cpu
      python tools/summarize_profiles.py ${TAG}_train k_gs_bin k_gs_accumulate k_ft_encode_levels k_field_train_forward k_field_train_backward k_composite_train k_march_train k_dg_ --sources train --dst $D --top 30 \
        --title "bench.py --mode train --steps 16 --warmup 4 --settle 64 (4,096-ray steps on the early, nearly full occupancy grid), 1x MI355X" > gpurun_out/sum_train.log 2>&1
      rm -rf gpurun_out/${TAG}_train_kt gpurun_out/${TAG}_train_pmc? ;;
    steady)
      # the steady state of training (grid converged, ~0.65 M points per step): kernel trace only, summarised over the last 24 steps + one step's timeline
      mkdir -p $D
      R=$(pwd); ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/${TAG}_steady_kt && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_steady_kt -- python3 $R/bench.py --mode train --steps 32 --warmup 4 --settle 1500 --no-cpu > $R/gpurun_out/${TAG}_steady_kt.log 2>&1 )
      { echo "# ${TAG}_train_steady: bench.py --mode train --steps 32 --warmup 4 --settle 1500 under rocprofv3 --kernel-trace, 1x MI355X; kernel sources $(python -c 'import bench; print(bench.sources_sha16(bench.TRAIN_SOURCES))')"
        echo; echo "The command's own line: \`$(grep '^{"metric"' gpurun_out/${TAG}_steady_kt.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.3f ms per step, %.2f M rays/s, %.0f points per step" % (d["ms_per_step"], d["value"]/1e6, d["config"]["points_per_step"]))')\`"
        echo; echo '## GPU time per step, last 24 steps (`tools/last_steps_kernel_stats.py`)'; echo; echo '```'
        python tools/last_steps_kernel_stats.py gpurun_out/${TAG}_steady_kt/*/*kernel_trace.csv 24 60
        echo '```'; echo; echo '## One step, launch by launch (start since the compositor, duration, gap to the previous launch; us)'; echo; echo '```'
        python tools/one_step_timeline.py gpurun_out/${TAG}_steady_kt/*/*kernel_trace.csv
        echo '```'; } > $D/${TAG}_train_steady_summary.md
      rm -rf gpurun_out/${TAG}_steady_kt ;;
    perop)
      bash tools/profile_round.sh ${TAG}_perop short -- bench.py --path per_op --no-cpu --no-fit --frames-per-launch 1 --steps 8 --warmup 2
      python tools/summarize_profiles.py ${TAG}_perop k_march_rays k_grid_forward k_ffmlp_forward k_composite_rays k_compact --dst $D --top 30 \
        --title "bench.py --path per_op (the reference-shaped op-by-op loop, 800x800, 64 iterations per frame), 1x MI355X" > gpurun_out/sum_perop.log 2>&1
      rm -rf gpurun_out/${TAG}_perop_kt gpurun_out/${TAG}_perop_pmc? ;;
    nav)
      bash tools/profile_round.sh ${TAG}_nav short -- tools/time_nav.py --only filter_native,filter_frozen,planner_graphed
      python tools/summarize_profiles.py ${TAG}_nav k_nav_run_bwd k_nav_run_fwd k_nav_density --dst $D --top 30 \
        --title "tools/time_nav.py: pose-filter iteration (run() 1,024 rays x 512 steps + backward) native and op-chain, planner query op-chain, 1x MI355X" > gpurun_out/sum_nav.log 2>&1
      rm -rf gpurun_out/${TAG}_nav_kt gpurun_out/${TAG}_nav_pmc? ;;
    dropin)
      # the unchanged-caller loop (nerf/renderer.py:325-374 over the drop-in ops): kernel trace of 3 frames, the last one accounted launch by launch
      mkdir -p $D
      R=$(pwd); ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/${TAG}_dropin_kt && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_dropin_kt -- python3 $R/bench.py --path drop_in --steps 3 --warmup 2 --no-cpu --no-fit --frames-per-launch 1 > $R/gpurun_out/${TAG}_dropin_kt.log 2>&1 )
      { echo "# ${TAG}_dropin: bench.py --path drop_in --steps 3 --warmup 2 under rocprofv3 --kernel-trace, 1x MI355X (800x800 S-ring frame, run_cuda as an unmodified renderer runs it)"
        echo; echo "The command's own line: \`$(grep '^{"metric"' gpurun_out/${TAG}_dropin_kt.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.2f ms per frame under the profiler" % d["ms_per_step"])')\`"
        echo; echo '```'
        python tools/drop_in_timeline.py gpurun_out/${TAG}_dropin_kt/*/*kernel_trace.csv 1 2
        echo '```'; } > $D/${TAG}_dropin_summary.md
      rm -rf gpurun_out/${TAG}_dropin_kt ;;
    trained)
      bash tools/profile_round.sh ${TAG}_trained full -- bench.py --model trained --fit-steps 2000 --steps 10 --warmup 2 --no-cpu --no-nav-block --no-drop-in-block --frames-per-launch 1
      python tools/summarize_profiles.py ${TAG}_trained k_render_frame --last 5 --dst $D --top 12 \
        --title "bench.py --model trained --fit-steps 2000 (student fitted to the hand-set scene), 800x800, 1x MI355X" > gpurun_out/sum_trained.log 2>&1
      rm -rf gpurun_out/${TAG}_trained_kt gpurun_out/${TAG}_trained_pmc? ;;
  esac
done
du -sh gpurun_out
