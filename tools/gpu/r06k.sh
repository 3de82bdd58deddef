# round 6: the weight-gradient GEMMs' token-dimension splits IN the step (E2K_WGRAD_SPLITS: 0 = the library's alone-timed cost model, 1 / 2 / 4 fixed),
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
F="--no-cpu-baseline --no-optimizer-leg --no-launch-floor --no-warm-leg"
for i in 1 2; do
for v in 0 2 3 4 8; do
  (E2K_WGRAD_SPLITS_DUAL=$v timeout 400 python bench.py $F --steps 20 --warmup 5) > gpurun_out/r06l_cfg3_${v}_${i}.log 2>&1
  python - $v $i <<'P'
import json, sys
for l in open(f'gpurun_out/r06l_cfg3_{sys.argv[1]}_{sys.argv[2]}.log'):
    if l.startswith('{'):
        d = json.loads(l); g = d['kernel_groups_ms_per_step']
        print('cfg3 E2K_WGRAD_SPLITS_DUAL=' + sys.argv[1], 'run', sys.argv[2], round(d['ms_per_step'], 3), 'tn_group', g.get('gemm_tn_group_bf16'), 'tn_dual', g.get('gemm_tn_dual_bf16'))
P
done; done
