# round 6, final call: the driver's test command on the final tree, the driver's bench command, single-stream rocprofv3 kernel stats,
# NT launch-mix shapes + HBM traffic counters, sample() at cfg5
#   gpurun --timeout 2700 -- 'bash tools/gpu/r06z.sh'
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r06z
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 300 python -c 'import __graft_entry__ as g; g.smoke()') > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log | cut -c1-200)"
(timeout 1800 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$? $(tail -1 $O/pytest_gpu.log)"
(timeout 900 python bench.py --steps 20 --warmup 5) > $O/bench_default.log 2>&1; echo "bench rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_default.log | head -1)"
(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor --dump-ops $O/ops_by_shape.json) > $O/bench_ops.log 2>&1
(timeout 200 python tools/nt_shapes.py) > $O/nt_shapes.log 2>&1; cp gpurun_out/nt_shapes.json $O/nt_shapes.json; cp gpurun_out/nt_shapes.json tools/nt_shapes_cfg3.json; tail -1 $O/nt_shapes.log
(timeout 300 python tools/bench_sample.py) > $O/sample.log 2>&1; tail -2 $O/sample.log
(timeout 300 python bench.py --config cfg2 --steps 40 --warmup 10 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_cfg2.log 2>&1; echo "cfg2 $(grep -o '"ms_per_step": [0-9.]*' $O/bench_cfg2.log | head -1)"
cd /tmp && export TMPDIR=/tmp
(E2K_LANES=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ss -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor --no-optimizer-leg --no-warm-leg) > $O/prof_ss.log 2>&1; echo "prof rc=$?"
find /tmp/prof_ss -name "*kernel_stats.csv" -exec cp {} $O/prof_ss_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/nt_traffic_probe.py) > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  find /tmp/pmc_$c -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/nt_traffic_$c.csv \;
done
head -6 $O/prof_ss_kernel_stats.csv | cut -c1-160
