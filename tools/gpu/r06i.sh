# round 6: e2k_hc_fwd_norm on hardware -- tests, then sample() at cfg5 with the fusion off / on (E2K_FUSE_HC_NORM 0 / 1), then the training step
# with 0 / 2, interleaved on one box
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for i in 1 2; do
for v in nosplit split; do
  fl=0; [ $v = split ] && fl=512
  (E2K_GEMM_FLAGS=$fl timeout 300 python tools/bench_sample.py) > gpurun_out/r06i_sample_${v}_${i}.log 2>&1
  echo "sample $v run $i: $(grep -o '"seconds": [0-9.]*' gpurun_out/r06i_sample_${v}_${i}.log) $(grep -o '"model_tflops_per_s": [0-9.]*' gpurun_out/r06i_sample_${v}_${i}.log)"
done; done
F="--no-cpu-baseline --no-optimizer-leg --no-launch-floor --no-warm-leg"
for i in 1 2; do
for v in 0 2; do
  (E2K_FUSE_HC_NORM=$v timeout 400 python bench.py $F --steps 20 --warmup 5) > gpurun_out/r06i_cfg3_${v}_${i}.log 2>&1
  (E2K_FUSE_HC_NORM=$v timeout 400 python bench.py $F --config cfg2 --steps 40 --warmup 10) > gpurun_out/r06i_cfg2_${v}_${i}.log 2>&1
  python - $v $i <<'P'
import json, sys
for c in ('cfg3', 'cfg2'):
    for l in open(f'gpurun_out/r06i_{c}_{sys.argv[1]}_{sys.argv[2]}.log'):
        if l.startswith('{'):
            d = json.loads(l); g = d['kernel_groups_ms_per_step']
            print(c, 'E2K_FUSE_HC_NORM=' + sys.argv[1], 'run', sys.argv[2], round(d['ms_per_step'], 3), 'launches', d['launches_per_step'], 'hc_fwd', g.get('hc_fwd'), g.get('hc_fwd_norm'), 'rmsnorm_fwd', g.get('rmsnorm_fwd'))
P
done; done
