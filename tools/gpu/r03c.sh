# Round 3, GPU call 3: where the per-tile overhead of the big NT tiles goes (store-burst probes) and why the forced data-parallel
# exchange costs 13 ms at world size 1 (RCCL channel count / bucket size / deferred exchange A/Bs)
tag=${1:-r03c}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python tools/probes/nt_store_burst.py) > gpurun_out/nt_store_burst_$tag.log 2>&1; echo "burst probe rc=$?"; grep -v amdgpu.ids gpurun_out/nt_store_burst_$tag.log | cut -c1-700 | tail -n 6
(timeout 300 python -m pytest tests/test_optim.py tests/test_kernels_gemm.py -m gpu -q -p no:cacheprovider -x -k 'adopt or persistent') > gpurun_out/pytest_${tag}_new.log 2>&1; echo "pytest new rc=$?"; tail -n 3 gpurun_out/pytest_${tag}_new.log
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -n 2; }
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor"
run plan $B
run ddp_bf16 $B --force-ddp
run ddp_bf16_2ch env NCCL_MAX_NCHANNELS=2 NCCL_MIN_NCHANNELS=2 $B --force-ddp
run ddp_bf16_8ch env NCCL_MAX_NCHANNELS=8 $B --force-ddp
run ddp_bf16_b6 $B --force-ddp --bucket-layers 6
run ddp_bf16_defer $B --force-ddp --ddp-defer
run ddp_fp32_defer $B --force-ddp --ddp-defer --grad-dtype fp32
