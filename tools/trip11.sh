mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python -m pytest tests/test_kernels_hc.py tests/test_e2tts.py -m gpu -q -p no:cacheprovider -x) > gpurun_out/pytest_gpu11.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/pytest_gpu11.log
(timeout 100 python tools/microbench.py hc) 2>&1 | grep -E "hc_"
(timeout 400 python bench.py --steps 4 --warmup 3 --no-cpu-baseline) > gpurun_out/bench11.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench11.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
(timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof11 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-graphs) > $GRAFT_REPO_ROOT/gpurun_out/prof11.log 2>&1; echo "prof rc=$?"
find /tmp/prof11 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof11_kernel_stats.csv \;
head -n 40 $GRAFT_REPO_ROOT/gpurun_out/prof11_kernel_stats.csv | cut -c1-200
