# quick A/B of bench variants on one box:  gpurun --timeout 900 -- 'bash tools/gpu/quick.sh'
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
for v in "" "--force-ddp"; do
  (timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $v) > gpurun_out/bench_q.log 2>&1; echo "[$v] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_q.log) $(grep -o '"launch_mode": "[^"]*"' gpurun_out/bench_q.log)"
done
