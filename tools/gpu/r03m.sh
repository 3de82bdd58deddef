# Round 3, GPU call 13: stream priorities for the launch lanes (side lanes low / chain high), A/B
tag=${1:-r03m}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
python - <<'P'
import torch
print('priority range', torch.cuda.Stream.priority_range())
P
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | tail -n 2; }
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor"
run base $B
run lanes_low env E2K_LANE_PRIORITY=low $B
run base2 $B
run lanes_low2 env E2K_LANE_PRIORITY=low $B
