# round 6: GEGLU as the first feed-forward GEMM's epilogue in TRAINING passes too (pre-activation stored), E2K_FUSE_GEGLU_TRAIN 0 / 1, interleaved;
# backbone / e2tts GPU tests on the tree with the dual-source weight-gradient launches at two splits
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests/test_backbone.py tests/test_e2tts.py tests/test_optim.py -m gpu -q -p no:cacheprovider -x) 2>&1 | tail -3
F="--no-cpu-baseline --no-optimizer-leg --no-launch-floor --no-warm-leg"
for i in 1 2 3; do
for v in 0 1; do
  (E2K_FUSE_GEGLU_TRAIN=$v timeout 400 python bench.py $F --steps 20 --warmup 5) > gpurun_out/r06n_cfg3_${v}_${i}.log 2>&1
  python - $v $i <<'P'
import json, sys
for l in open(f'gpurun_out/r06n_cfg3_{sys.argv[1]}_{sys.argv[2]}.log'):
    if l.startswith('{'):
        d = json.loads(l); g = d['kernel_groups_ms_per_step']
        print('cfg3 E2K_FUSE_GEGLU_TRAIN=' + sys.argv[1], 'run', sys.argv[2], round(d['ms_per_step'], 3), 'launches', d['launches_per_step'], 'nt', g.get('gemm_nt_bf16'), 'nt_geglu', g.get('gemm_nt_geglu_bf16'), 'geglu_fwd', g.get('geglu_fwd'))
P
done; done
