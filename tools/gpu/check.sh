# Full GPU check on a gpurun box:  gpurun --timeout 1500 -- 'bash tools/gpu/check.sh [tag]'
#   pytest -m gpu, bench runs (plan replay / eager A/B), and a rocprofv3 kernel-stats profile of a bench run (-> gpurun_out/)
tag=${1:-check}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=8) > gpurun_out/pytest_$tag.log 2>&1; echo "pytest rc=$?"; tail -n 14 gpurun_out/pytest_$tag.log
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log) $(grep -o '"frac": [0-9.]*' gpurun_out/bench_${tag}_$t.log | head -1) $(grep -o '"launches_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log | head -1)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | tail -n 2; }
run plan python bench.py --steps 8 --warmup 2
run plan_t256 env E2K_GEMM_FLAGS=256 python bench.py --steps 8 --warmup 2 --no-cpu-baseline
run plan_fuse_geglu env E2K_FUSE_GEGLU=1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline     # A/B: GEGLU as GEMM1's epilogue (off by default until timed)
run eager python bench.py --steps 6 --warmup 2 --no-cpu-baseline --eager
cd /tmp && export TMPDIR=/tmp
(timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor) > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1; echo "prof rc=$?"
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_kernel_stats.csv \;
head -n 14 $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_kernel_stats.csv | cut -c1-150
