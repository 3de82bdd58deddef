# Round 3, GPU call 5: staged NT epilogue with 16-byte stores + LDS-transposing weight-gradient reduce (per-shape A/B, bench A/B),
# bisect of the forced data-parallel overhead, MelSpec banded filterbank timing
tag=${1:-r03e}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 400 python -m pytest tests/test_kernels_gemm.py tests/test_e2tts.py -m gpu -q -p no:cacheprovider -x -k 'gemm or melspec') > gpurun_out/pytest_${tag}_gemm.log 2>&1; echo "pytest gemm+melspec rc=$?"; tail -n 3 gpurun_out/pytest_${tag}_gemm.log
(timeout 300 python tools/probes/gemm_epilogue_ab.py) > gpurun_out/gemm_epilogue_ab_$tag.log 2>&1; echo "epilogue probe rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_epilogue_ab_$tag.log | cut -c1-330 | tail -n 26
(timeout 120 python tools/bench_melspec.py) > gpurun_out/melspec_$tag.log 2>&1; echo "melspec rc=$?"; grep -v amdgpu.ids gpurun_out/melspec_$tag.log | tail -n 3 | cut -c1-300
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log) $(grep -o '"frac": [0-9.]*' gpurun_out/bench_${tag}_$t.log | head -1) $(grep -o '"gemm_tn_bf16": {[^}]*}' gpurun_out/bench_${tag}_$t.log | head -1) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -n 2; }
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor"
run plan $B
run no_stage env E2K_GEMM_FLAGS=64 $B
run bisect_init $B --ddp-bisect init
run bisect_nohook $B --ddp-bisect nohook
run bisect_nooutside $B --ddp-bisect nooutside --grad-dtype fp32
run ddp_fp32 $B --force-ddp --grad-dtype fp32
run plan_again $B
