mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 400 python -m pytest tests/test_backbone.py -m gpu -q -p no:cacheprovider -x) > gpurun_out/pytest_gpu10.log 2>&1; echo "pytest rc=$?"; tail -n 25 gpurun_out/pytest_gpu10.log
(timeout 400 python bench.py --steps 4 --warmup 3 --no-cpu-baseline) > gpurun_out/bench10.log 2>&1; echo "bench rc=$?"; tail -n 3 gpurun_out/bench10.log | cut -c1-1500
