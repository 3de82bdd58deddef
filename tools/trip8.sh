mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider) > gpurun_out/pytest_gpu8.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/pytest_gpu8.log
(timeout 200 python tools/microbench.py tn attn) > gpurun_out/microbench8.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu gpurun_out/microbench8.log | grep -E "use_tr=1|attn"
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof8 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline) > $GRAFT_REPO_ROOT/gpurun_out/rocprof8.log 2>&1; echo "rocprof rc=$?"
cp /tmp/prof8/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r8_kernel_stats.csv
tail -n 1 $GRAFT_REPO_ROOT/gpurun_out/rocprof8.log | cut -c1-300
