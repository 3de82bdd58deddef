# round 6: remainder split for the fused GEGLU-backward GEMM only (E2K_GEGLU_BWD_SPLIT 0 / 1), interleaved; + the 8-interval cfg5-exact case
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
F="--no-cpu-baseline --no-optimizer-leg --no-launch-floor --no-warm-leg"
for i in 1 2 3; do
for v in 0 1; do
  (E2K_GEGLU_BWD_SPLIT=$v timeout 400 python bench.py $F --steps 20 --warmup 5) > gpurun_out/r06j_cfg3_${v}_${i}.log 2>&1
  python - $v $i <<'P'
import json, sys
for l in open(f'gpurun_out/r06j_cfg3_{sys.argv[1]}_{sys.argv[2]}.log'):
    if l.startswith('{'):
        d = json.loads(l); g = d['kernel_groups_ms_per_step']
        print('cfg3 E2K_GEGLU_BWD_SPLIT=' + sys.argv[1], 'run', sys.argv[2], round(d['ms_per_step'], 3), 'geglu_bwd gemm', g.get('gemm_nt_geglu_bwd_bf16'))
P
done; done
(timeout 900 python -m pytest tests/test_fullsize.py -m gpu -q -p no:cacheprovider -k "cfg5_exact") 2>&1 | tail -3
