mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python __graft_entry__.py smoke) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
(timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -n 30 gpurun_out/pytest_gpu.log
(timeout 400 python tools/microbench.py) > gpurun_out/microbench.log 2>&1; echo "microbench rc=$?"
(timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline) > gpurun_out/bench_cfg3.log 2>&1; echo "bench rc=$?"
tail -n 5 gpurun_out/bench_cfg3.log
tail -n 3 gpurun_out/smoke.log
