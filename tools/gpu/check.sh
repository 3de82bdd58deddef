# Full GPU check on a gpurun box:  gpurun --timeout 1500 -- 'bash tools/gpu/check.sh [tag]'
#   pytest -m gpu, a bench run, and a rocprofv3 kernel-stats profile of an eager bench run (-> gpurun_out/)
tag=${1:-check}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -x) > gpurun_out/pytest_$tag.log 2>&1; echo "pytest rc=$?"; tail -n 2 gpurun_out/pytest_$tag.log
(timeout 400 python bench.py --steps 6 --warmup 3) > gpurun_out/bench_$tag.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench_$tag.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
(timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-graphs) > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1; echo "prof rc=$?"
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_kernel_stats.csv \;
head -n 12 $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_kernel_stats.csv | cut -c1-150
