# round 6: runtime environment A/B on the no-split tree: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory) 0 / 1, cfg2 + cfg3,
# interleaved; then the split GPU test
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
F="--no-cpu-baseline --no-optimizer-leg --no-launch-floor --no-warm-leg"
for i in 1 2; do
for v in 0 1; do
  (HIP_FORCE_DEV_KERNARG=$v timeout 400 python bench.py $F --steps 20 --warmup 5) > gpurun_out/r06g_cfg3_ka${v}_$i.log 2>&1
  (HIP_FORCE_DEV_KERNARG=$v timeout 400 python bench.py $F --config cfg2 --steps 40 --warmup 10) > gpurun_out/r06g_cfg2_ka${v}_$i.log 2>&1
  python - $v $i <<'P'
import json, sys
for c in ('cfg3', 'cfg2'):
    for l in open(f'gpurun_out/r06g_{c}_ka{sys.argv[1]}_{sys.argv[2]}.log'):
        if l.startswith('{'):
            d = json.loads(l); print(c, 'HIP_FORCE_DEV_KERNARG=' + sys.argv[1], 'run', sys.argv[2], round(d['ms_per_step'], 3), 'host', round(d['host_enqueue_ms_per_step'], 2))
P
done; done
(timeout 600 python -m pytest tests/test_kernels_gemm.py -m gpu -q -p no:cacheprovider -k "split") 2>&1 | tail -3
