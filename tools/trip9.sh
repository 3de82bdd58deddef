mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider) > gpurun_out/pytest_gpu9.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/pytest_gpu9.log
(timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline) > gpurun_out/bench9.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench9.log | cut -c1-200; grep -o '"host_enqueue_ms_per_step": [0-9.]*' gpurun_out/bench9.log
(timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-ddp) > gpurun_out/bench9_ddp.log 2>&1; echo "bench ddp rc=$?"; tail -n 2 gpurun_out/bench9_ddp.log | cut -c1-200
(timeout 300 python tools/bench_sample.py 32 32) > gpurun_out/sample9.log 2>&1; echo "sample rc=$?"; tail -n 2 gpurun_out/sample9.log
(timeout 100 python tools/microbench.py ew) 2>&1 | grep -E "dwconv"
