set -u
D=gpurun_out/profiles_r14
mkdir -p $D
bash tools/profile_round.sh r14 full -- bench.py --steps 30 --warmup 5 --no-cpu 2>&1 | tail -1
python tools/summarize_profiles.py r14 k_render_frame --dst $D --top 12 > /dev/null
bash tools/profile_round.sh r14_train short -- bench.py --mode train --steps 16 --warmup 4 --settle 64 2>&1 | tail -1
python tools/summarize_profiles.py r14_train k_grid_backward k_ffmlp_bwd_wgrad k_ffmlp_bwd_act k_ffmlp_forward k_grid_forward k_march_train_count k_composite_train k_dg_ --sources train --dst $D --top 40 --title "bench.py --mode train --steps 16 --warmup 4 --settle 64 (4,096-ray steps on the early, nearly full occupancy grid), 1x MI355X" > /dev/null
bash tools/profile_round.sh r14_perop short -- bench.py --steps 6 --warmup 2 --no-cpu --path per_op 2>&1 | tail -1
python tools/summarize_profiles.py r14_perop k_march_rays k_grid_forward k_ffmlp_forward k_composite_rays k_compact k_rm_build_coarse --dst $D --top 30 --title "bench.py --path per_op (the reference-shaped op-by-op loop, 800x800, 64 iterations per frame), 1x MI355X" > /dev/null
bash tools/profile_round.sh r14_nav short -- tools/time_nav.py --only filter_native,filter_frozen,planner_frozen 2>&1 | tail -1
python tools/summarize_profiles.py r14_nav k_nav_run_bwd k_nav_run_fwd k_grid_input_backward_recompute k_grid_forward --dst $D --top 30 --title "tools/time_nav.py: pose-filter iteration (run() 1,024 rays x 512 steps + backward) native and op-chain, planner query op-chain, 1x MI355X" > /dev/null
cp gpurun_out/r14*_bench_under_profiler.json gpurun_out/r14*_sources.txt $D/ 2>/dev/null
rm -rf gpurun_out/r14*_pmc* gpurun_out/r14*_kt gpurun_out/r14*.log
ls $D; du -sh gpurun_out
