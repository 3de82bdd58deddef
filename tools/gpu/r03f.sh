# Round 3, GPU call 6: slab-parallel weight-gradient reduce (per-shape + bench), cfg5 sample(), process-group overhead knobs
tag=${1:-r03f}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python -m pytest tests/test_kernels_gemm.py -m gpu -q -p no:cacheprovider -x -k 'tn') > gpurun_out/pytest_${tag}_gemm.log 2>&1; echo "pytest tn rc=$?"; tail -n 3 gpurun_out/pytest_${tag}_gemm.log
(timeout 300 python tools/probes/gemm_epilogue_ab.py) > gpurun_out/gemm_epilogue_ab_$tag.log 2>&1; echo "epilogue probe rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_epilogue_ab_$tag.log | cut -c1-200 | tail -n 12
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log) $(grep -o '"frac": [0-9.]*' gpurun_out/bench_${tag}_$t.log | head -1) $(grep -o '"gemm_tn_bf16": {[^}]*}' gpurun_out/bench_${tag}_$t.log | head -1) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl\|destroy_process" | tail -n 2; }
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor"
run plan $B
run bisect_init $B --ddp-bisect init
run bisect_init_lazy env E2K_BENCH_LAZY_PG=1 $B --ddp-bisect init
run bisect_init_nomon env TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0 $B --ddp-bisect init
run plan_again $B
(timeout 300 python tools/bench_sample.py 32 32) > gpurun_out/sample_$tag.log 2>&1; echo "sample rc=$?"; grep -v amdgpu.ids gpurun_out/sample_$tag.log | tail -n 2 | cut -c1-400
