# round 6: attn_bwd_prep folded into the dQ kernel's prologue (-48 launches per step): attention tests on hardware, then same-box A/B of the
# step against the library of the previous commit (tools/ab/lib/libe2k_head.so), interleaved, cfg3 + cfg2; HSA_ENABLE_INTERRUPT=0 as a third arm
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 600 python -m pytest tests/test_kernels_attn.py tests/test_backbone.py -m gpu -q -p no:cacheprovider -x) 2>&1 | tail -3
F="--no-cpu-baseline --no-optimizer-leg --no-launch-floor --no-warm-leg"
for i in 1 2; do
for v in head new newpoll; do
  unset E2K_LIB HSA_ENABLE_INTERRUPT
  [ $v = head ] && export E2K_LIB=$PWD/tools/ab/lib/libe2k_head.so
  [ $v = newpoll ] && export HSA_ENABLE_INTERRUPT=0
  (timeout 400 python bench.py $F --steps 20 --warmup 5) > gpurun_out/r06h_cfg3_${v}_$i.log 2>&1
  (timeout 400 python bench.py $F --config cfg2 --steps 40 --warmup 10) > gpurun_out/r06h_cfg2_${v}_$i.log 2>&1
  python - $v $i <<'P'
import json, sys
for c in ('cfg3', 'cfg2'):
    for l in open(f'gpurun_out/r06h_{c}_{sys.argv[1]}_{sys.argv[2]}.log'):
        if l.startswith('{'):
            d = json.loads(l); g = d['kernel_groups_ms_per_step']
            print(c, sys.argv[1], 'run', sys.argv[2], round(d['ms_per_step'], 3), 'launches', d['launches_per_step'], 'attn_bwd', g.get('attn_bwd'))
P
done; done
