mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -x) > gpurun_out/pytest_gpu23.log 2>&1; echo "pytest rc=$?"; tail -n 2 gpurun_out/pytest_gpu23.log
(timeout 400 python bench.py --steps 4 --warmup 3 --no-cpu-baseline) > gpurun_out/bench23.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench23.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
(timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof23 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-graphs) > $GRAFT_REPO_ROOT/gpurun_out/prof23.log 2>&1; echo "prof rc=$?"
find /tmp/prof23 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof23_kernel_stats.csv \;
head -n 30 $GRAFT_REPO_ROOT/gpurun_out/prof23_kernel_stats.csv | cut -c1-150
