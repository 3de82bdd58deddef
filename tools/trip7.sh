mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python -m pytest tests/test_kernels_gemm.py tests/test_e2tts.py -m gpu -q -p no:cacheprovider) > gpurun_out/pytest_gpu7.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/pytest_gpu7.log
(timeout 200 python tools/microbench.py gemm) > gpurun_out/microbench7.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu gpurun_out/microbench7.log | grep -E "gemm_nt"
(timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline) > gpurun_out/bench7.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench7.log | cut -c1-300
