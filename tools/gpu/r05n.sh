# round 5: where sample() goes at cfg5 (kernel stats of 4 midpoint steps at B 32), cfg2 bench line
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05n
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 300 python bench.py --config cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/bench_cfg2.log 2>&1; echo "cfg2 rc=$?"; grep -o '"ms_per_step": [0-9.]*' $O/bench_cfg2.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sample -- python $GRAFT_REPO_ROOT/tools/bench_sample.py 32 4) > $O/prof_sample.log 2>&1; echo "prof rc=$?"; tail -1 $O/prof_sample.log | cut -c1-300
find /tmp/prof_sample -name "*kernel_stats.csv" -exec cp {} $O/sample_kernel_stats.csv \;
head -25 $O/sample_kernel_stats.csv | cut -c1-200
