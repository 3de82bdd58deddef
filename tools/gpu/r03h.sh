# Round 3, GPU call 8: hc_bwd two-kernel reproducer (round-2 kernel with LDS float atomics vs today's), the whole -m gpu suite (with the
# 200-trial co-residency screen), dual-source weight-gradient A/B
tag=${1:-r03h}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 200 python tools/probes/hc_bwd_repro/run.py) > gpurun_out/hc_bwd_repro_$tag.log 2>&1; echo "hc repro rc=$?"; grep -v amdgpu.ids gpurun_out/hc_bwd_repro_$tag.log | tail -n 4 | cut -c1-300
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log) $(grep -o '"gemm_tn_bf16": {[^}]*}' gpurun_out/bench_${tag}_$t.log | head -1) $(grep -o '"gemm_tn_dual_bf16": {[^}]*}' gpurun_out/bench_${tag}_$t.log | head -1) $(grep -o '"launches_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log | head -1)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | tail -n 2; }
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor"
run dual $B
run no_dual env E2K_WGRAD_DUAL=0 $B
run dual_again $B
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6) > gpurun_out/pytest_$tag.log 2>&1; echo "pytest rc=$?"; tail -n 14 gpurun_out/pytest_$tag.log
