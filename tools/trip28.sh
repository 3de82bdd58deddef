mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 250 python tools/find_syncs.py) 2>&1 | grep -E "SYNC|host:"
(timeout 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline) > gpurun_out/bench28.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench28.log | cut -c1-330; grep -o '"host_enqueue_ms_per_step": [0-9.]*' gpurun_out/bench28.log
