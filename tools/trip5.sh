mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider) > gpurun_out/pytest_gpu5.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/pytest_gpu5.log
(timeout 200 python tools/microbench.py gemm ew) > gpurun_out/microbench5.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu gpurun_out/microbench5.log | grep -E "gemm_nt|dwconv"
(timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline) > gpurun_out/bench5.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench5.log | cut -c1-300
