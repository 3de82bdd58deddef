mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider) > gpurun_out/pytest_gpu4.log 2>&1; echo "pytest rc=$?"; tail -n 5 gpurun_out/pytest_gpu4.log
(timeout 200 python tools/microbench.py attn hc) > gpurun_out/microbench4.log 2>&1; echo "microbench rc=$?"; grep -v amdgpu gpurun_out/microbench4.log
(timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline) > gpurun_out/bench4.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench4.log | cut -c1-300
