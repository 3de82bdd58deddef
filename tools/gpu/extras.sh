# extra measurements for the docs: sampler throughput (eager / graphs) and the kernel micro-benchmarks
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
(timeout 200 python tools/bench_sample.py 32 32) 2>&1 | tail -1 | cut -c1-300
(timeout 200 python tools/bench_sample.py 32 32 --graphs) 2>&1 | tail -1 | cut -c1-300
(timeout 300 python tools/microbench.py) > gpurun_out/microbench_all.log 2>&1; echo "microbench rc=$?"; tail -n 60 gpurun_out/microbench_all.log | grep -c name
