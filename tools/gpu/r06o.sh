# round 6: bf16 weight shadows cast layer by layer on the WGRAD lane ahead of their use (E2K_RECAST_W_ON_LANE 0 / 1), interleaved; test first
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests/test_backbone.py tests/test_optim.py -m gpu -q -p no:cacheprovider -x -k "recast or plan or lanes or optim or adopt or training") 2>&1 | tail -3
F="--no-cpu-baseline --no-launch-floor --no-warm-leg"
for i in 1 2 3; do
for v in 0 1; do
  (E2K_RECAST_W_ON_LANE=$v timeout 400 python bench.py $F --steps 20 --warmup 5) > gpurun_out/r06o_cfg3_${v}_${i}.log 2>&1
  python - $v $i <<'P'
import json, sys
for l in open(f'gpurun_out/r06o_cfg3_{sys.argv[1]}_{sys.argv[2]}.log'):
    if l.startswith('{'):
        d = json.loads(l); o = d.get('optimizer_leg') or {}
        print('cfg3 E2K_RECAST_W_ON_LANE=' + sys.argv[1], 'run', sys.argv[2], round(d['ms_per_step'], 3), 'launches', d['launches_per_step'], 'with clip+adopt', round(o.get('ms_per_step_with_clip_adopt', 0), 2), 'plain in that loop', round(o.get('ms_per_step_fwd_bwd', 0), 2))
P
done; done
