# First GPU call of round 3 (≈ 20 box-minutes):  gpurun --timeout 1500 -- 'bash tools/gpu/round3_first.sh'
#   1. pytest -m gpu, then the GEGLU-epilogue tests that have only run on the kernel model so far (E2K_TEST_UNTIMED=1)
#   2. the in-kernel weight-gradient finish probe and the fused-GEGLU probe (bit comparison + per-shape timing against the two launches each replaces)
#   3. bench A/Bs: default | E2K_FUSE_GEGLU=1 | E2K_TN_SELF_REDUCE=1 | E2K_GEMM_FLAGS=512 (in-kernel NT fix-up) | both | E2K_LANES=0 (single stream; lane_ms_per_step of the default line names the critical lane)
#   4. rocprofv3 kernel stats on one stream (the durations roofline.avg_launch_ms must agree with)
tag=${1:-r03a}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8) > gpurun_out/pytest_$tag.log 2>&1; echo "pytest rc=$?"; tail -n 14 gpurun_out/pytest_$tag.log
# the hardware variants of the tests of the never-executed E2K_FUSE_GEGLU path, on their own and under a short timeout
(E2K_TEST_UNTIMED=1 timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k 'geglu or gelu_erf or self_reduce or self_fixup') > gpurun_out/pytest_${tag}_geglu.log 2>&1; echo "pytest geglu rc=$?"; tail -n 6 gpurun_out/pytest_${tag}_geglu.log
(timeout 300 python tools/probes/tn_self_reduce.py) > gpurun_out/tn_self_reduce_$tag.log 2>&1; echo "tn self-reduce probe rc=$?"; tail -n 12 gpurun_out/tn_self_reduce_$tag.log | cut -c1-300
(timeout 300 python tools/probes/geglu_fused.py) > gpurun_out/geglu_fused_$tag.log 2>&1; echo "geglu probe rc=$?"; tail -n 4 gpurun_out/geglu_fused_$tag.log | cut -c1-400
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log) $(grep -o '"lane_ms_per_step": {[^}]*}' gpurun_out/bench_${tag}_$t.log) $(grep -o '"frac": [0-9.]*' gpurun_out/bench_${tag}_$t.log | head -1)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | tail -n 2; }
run plan python bench.py --steps 8 --warmup 2
run fuse_geglu env E2K_FUSE_GEGLU=1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline
run tn_self_reduce env E2K_TN_SELF_REDUCE=1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline
run nt_self_fixup env E2K_GEMM_FLAGS=512 python bench.py --steps 8 --warmup 2 --no-cpu-baseline
run all_in_kernel env E2K_GEMM_FLAGS=512 E2K_TN_SELF_REDUCE=1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline
run single_stream env E2K_LANES=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline
run plan_again python bench.py --steps 8 --warmup 2 --no-cpu-baseline
bash tools/gpu/prof_single_stream.sh
