# round 6, after the final call: rocprofv3 kernel stats of the final tree WITH the launch lanes (the durations stretch where lanes overlap;
# the single-stream summary of r06z.sh is the one roofline.avg_launch_ms agrees with)
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r06z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lanes -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor --no-optimizer-leg --no-warm-leg) > $O/prof_lanes.log 2>&1; echo "prof rc=$?"
find /tmp/prof_lanes -name "*kernel_stats.csv" -exec cp {} $O/prof_lanes_kernel_stats.csv \;
head -8 $O/prof_lanes_kernel_stats.csv | cut -c1-170
