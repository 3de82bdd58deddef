# Round 3, GPU call 7: does an RCCL communicator cost 11 ms because the process runs out of hardware queues?  (GPU_MAX_HW_QUEUES);
# staged GEGLU epilogue in sample(); kernel-trace idle analysis of a step
tag=${1:-r03g}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python -m pytest tests/test_kernels_gemm.py tests/test_backbone.py -m gpu -q -p no:cacheprovider -x -k 'geglu') > gpurun_out/pytest_${tag}_geglu.log 2>&1; echo "pytest geglu rc=$?"; tail -n 3 gpurun_out/pytest_${tag}_geglu.log
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl\|destroy_process\|socket.cpp" | tail -n 2; }
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor"
run plan $B
run plan_q8 env GPU_MAX_HW_QUEUES=8 $B
run plan_q2 env GPU_MAX_HW_QUEUES=2 $B
run init $B --ddp-bisect init
run init_q8 env GPU_MAX_HW_QUEUES=8 $B --ddp-bisect init
run init_q16 env GPU_MAX_HW_QUEUES=16 $B --ddp-bisect init
run ddp_q8 env GPU_MAX_HW_QUEUES=8 $B --force-ddp
run ddp_q8_fp32 env GPU_MAX_HW_QUEUES=8 $B --force-ddp --grad-dtype fp32
(timeout 300 python tools/bench_sample.py 32 32) > gpurun_out/sample_$tag.log 2>&1; echo "sample rc=$?"; grep -v amdgpu.ids gpurun_out/sample_$tag.log | tail -n 1 | cut -c1-300
