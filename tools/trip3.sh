mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python -m pytest tests/test_kernels_gemm.py tests/test_kernels_hc.py tests/test_backbone.py -m gpu -q -p no:cacheprovider) > gpurun_out/pytest_gpu3.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/pytest_gpu3.log
(timeout 200 python tools/microbench.py gemm tn hc) > gpurun_out/microbench3.log 2>&1; echo "microbench rc=$?"
(timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline) > gpurun_out/bench3.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench3.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline) > $GRAFT_REPO_ROOT/gpurun_out/rocprof3.log 2>&1; echo "rocprof rc=$?"
find /tmp/prof3 -name "*stats*.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/ \;
ls -la /tmp/prof3/* | head; ls -la $GRAFT_REPO_ROOT/gpurun_out | head -20
