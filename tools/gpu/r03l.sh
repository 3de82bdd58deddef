# Round 3, GPU call 12: parameter-gradient reductions off the chain (A/B), what tensor-library ops a replayed step still issues,
# CPU baseline at full depth, the whole -m gpu suite on the final kernels
tag=${1:-r03l}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log) $(grep -o '"lane_ms_per_step": {[^}]*}[^}]*}[^}]*}' gpurun_out/bench_${tag}_$t.log) $(grep -o '"launches_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log | head -1)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | tail -n 2; }
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor"
run defer $B
run no_defer env E2K_DEFER_REDUCES=0 $B
run defer_again $B
(timeout 200 python tools/probes/step_torch_ops.py) > gpurun_out/step_torch_ops_$tag.log 2>&1; echo "torch ops rc=$?"; grep -v amdgpu.ids gpurun_out/step_torch_ops_$tag.log | head -n 30 | cut -c1-150
(timeout 300 python tools/cpu_cfg3_full_depth.py) > gpurun_out/cpu_full_$tag.log 2>&1; echo "cpu full depth rc=$?"; tail -n 1 gpurun_out/cpu_full_$tag.log | cut -c1-300
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x) > gpurun_out/pytest_$tag.log 2>&1; echo "pytest rc=$?"; tail -n 6 gpurun_out/pytest_$tag.log
