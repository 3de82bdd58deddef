#!/bin/bash
# first hardware run of the 256 x 256 8-phase NT kernel: small shapes under a short timeout first (a hang must not
# cost the box), then the cfg3 A/B, then the end-to-end bench with the kernel enabled where it fills the chip
#   gpurun --timeout 900 -- 'bash tools/gpu/t256.sh'
mkdir -p gpurun_out
timeout 180 python tools/gemm_t256_check.py --quick > gpurun_out/t256_quick.log 2>&1 || { echo "quick check failed / timed out"; tail -5 gpurun_out/t256_quick.log; exit 1; }
timeout 400 python tools/gemm_t256_check.py > gpurun_out/t256_full.log 2>&1; tail -3 gpurun_out/t256_full.log
timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
E2K_GEMM_FLAGS=256 timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_t256auto.json 2> gpurun_out/bench_t256auto.err
tail -1 gpurun_out/bench_default.json | cut -c1-200; tail -1 gpurun_out/bench_t256auto.json | cut -c1-200
