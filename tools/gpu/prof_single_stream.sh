# rocprofv3 kernel stats of the bench on ONE stream (E2K_LANES=0): the per-kernel durations that `roofline.avg_launch_ms` (each call
# timed alone) must agree with; with the launch lanes the kernels of different lanes overlap and their durations stretch
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
(E2K_LANES=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ss -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $GRAFT_REPO_ROOT/gpurun_out/prof_ss.log 2>&1; echo "prof rc=$?"
find /tmp/prof_ss -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_ss_kernel_stats.csv \;
grep -o '"ms_per_step": [0-9.]*' $GRAFT_REPO_ROOT/gpurun_out/prof_ss.log | head -1
head -n 6 $GRAFT_REPO_ROOT/gpurun_out/prof_ss_kernel_stats.csv | cut -c1-140
