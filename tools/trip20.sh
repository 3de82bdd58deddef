mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python -m pytest tests/test_kernels_elementwise.py tests/test_backbone.py -m gpu -q -p no:cacheprovider -x) > gpurun_out/pytest_gpu20.log 2>&1; echo "pytest rc=$?"; tail -n 2 gpurun_out/pytest_gpu20.log
(timeout 200 python tools/microbench.py ew) 2>&1 | grep -E "dwconv|rmsnorm|gate|geglu|colsum"
(timeout 400 python bench.py --steps 4 --warmup 3 --no-cpu-baseline) > gpurun_out/bench20.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench20.log | cut -c1-300
