# Round 3, GPU call 2:  gpurun --timeout 1200 -- 'bash tools/gpu/r03b.sh'
#   persistent NT kernel: hardware tests + per-shape A/B probe + bench A/B; forced data-parallel exchange (new pack kernels); cfg2 bench line;
#   the new full-size parity cases (half-strength off-init, cfg3 B = 2)
tag=${1:-r03b}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python -m pytest tests/test_kernels_gemm.py -m gpu -q -p no:cacheprovider -x -k 'persistent or geglu') > gpurun_out/pytest_${tag}_gemm.log 2>&1; echo "pytest gemm rc=$?"; tail -n 4 gpurun_out/pytest_${tag}_gemm.log
(timeout 300 python tools/probes/nt_persist_ab.py) > gpurun_out/nt_persist_ab_$tag.log 2>&1; echo "persist probe rc=$?"; grep -v amdgpu.ids gpurun_out/nt_persist_ab_$tag.log | cut -c1-420 | tail -n 12
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log) $(grep -o '"lane_ms_per_step": {[^}]*}[^}]*}[^}]*}' gpurun_out/bench_${tag}_$t.log) $(grep -o '"frac": [0-9.]*' gpurun_out/bench_${tag}_$t.log | head -1)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | tail -n 2; }
run plan python bench.py --steps 8 --warmup 2 --no-cpu-baseline
run persist env E2K_GEMM_FLAGS=64 python bench.py --steps 8 --warmup 2 --no-cpu-baseline
run force_ddp_bf16 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor --force-ddp
run force_ddp_fp32 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor --force-ddp --grad-dtype fp32
run cfg2 python bench.py --config cfg2 --steps 20 --warmup 5 --no-cpu-baseline
run plan_again python bench.py --steps 8 --warmup 2 --no-cpu-baseline
(timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6) > gpurun_out/pytest_$tag.log 2>&1; echo "pytest rc=$?"; tail -n 12 gpurun_out/pytest_$tag.log
