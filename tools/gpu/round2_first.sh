# First GPU call of the next round: everything that was written after round 1's GPU minutes ran out, in the order
# "cheap and safe first".   gpurun --timeout 1500 -- 'bash tools/gpu/round2_first.sh'
#   1. pytest -m gpu (includes the new reference-golden tests and FusedAdopt with persistent gradients)
#   2. emu-only tests run against the real library (persistent gradients, edge inputs, shared dropout masks)
#   2b. every kernel call of a backbone step vs an fp32 reference on its own inputs (tools/insitu_check.py)
#   3. 256 x 256 NT kernel: small shapes under a timeout, then race screen + A/B on the cfg3 shapes
#   4. bench A/B: default | --persistent-grads | E2K_GEMM_FLAGS=256 | both
#   5. micro-benchmarks of the element-wise kernels (depthwise-conv split backward vs fused)
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
(timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider) > gpurun_out/r2_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/r2_pytest.log
(timeout 300 python tools/gpu_variants_of_emu_tests.py) > gpurun_out/r2_emu_only.log 2>&1; echo "emu-only tests on the GPU rc=$?"; tail -n 5 gpurun_out/r2_emu_only.log
(timeout 200 python tools/insitu_check.py --gpu --lam 0) > gpurun_out/r2_insitu.log 2>&1; (timeout 200 python tools/insitu_check.py --gpu --lam 1) >> gpurun_out/r2_insitu.log 2>&1; echo "in-situ kernel check rc=$?"; grep -E 'whole|attn_bwd dQ|hc_bwd dR' gpurun_out/r2_insitu.log
(timeout 120 python tools/gemm_t256_check.py --quick) > gpurun_out/r2_t256_quick.log 2>&1; rc=$?; echo "t256 quick rc=$rc"; tail -n 2 gpurun_out/r2_t256_quick.log
if [ $rc -eq 0 ]; then
  (timeout 300 python tools/gemm_t256_check.py) > gpurun_out/r2_t256_full.log 2>&1; echo "t256 full rc=$?"; tail -n 13 gpurun_out/r2_t256_full.log | cut -c1-260
fi
run() { tag=$1; shift; (timeout 300 "$@") > gpurun_out/r2_bench_$tag.log 2>&1; echo "[$tag] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2_bench_$tag.log) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' gpurun_out/r2_bench_$tag.log) $(grep -o '"frac": [0-9.]*' gpurun_out/r2_bench_$tag.log | head -1)"; }
run default python bench.py --steps 6 --warmup 3 --no-cpu-baseline
run persist python bench.py --steps 6 --warmup 3 --no-cpu-baseline --persistent-grads
if [ $rc -eq 0 ]; then
  run t256 env E2K_GEMM_FLAGS=256 python bench.py --steps 6 --warmup 3 --no-cpu-baseline
  run t256_persist env E2K_GEMM_FLAGS=256 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --persistent-grads
fi
(timeout 200 python tools/microbench.py ew) > gpurun_out/r2_micro_ew.log 2>&1; echo "microbench ew rc=$?"; grep -i dwconv gpurun_out/r2_micro_ew.log | cut -c1-160
