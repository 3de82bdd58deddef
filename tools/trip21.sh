mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python -m pytest tests/test_kernels_gemm.py -m gpu -q -p no:cacheprovider -x) > gpurun_out/pytest_gpu21.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/pytest_gpu21.log
timeout 300 python tools/gemm_bound.py
