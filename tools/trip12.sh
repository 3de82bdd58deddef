mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python -m pytest tests/test_kernels_hc.py tests/test_backbone.py -m gpu -q -p no:cacheprovider -x) > gpurun_out/pytest_gpu12.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/pytest_gpu12.log
(timeout 100 python tools/microbench.py hc) 2>&1 | grep -E "hc_"
(timeout 400 python bench.py --steps 4 --warmup 3 --no-cpu-baseline) > gpurun_out/bench12.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench12.log | cut -c1-300
