# round 6: the weight-gradient lane alone confined to a slice of the CUs (E2K_LANE_CUS=",first:count"); round 4 only masked the TEXT lane
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
F="--no-cpu-baseline --no-launch-floor --no-warm-leg --no-optimizer-leg"
for i in 1 2; do
for v in none 0:192 0:128 0:64; do
  if [ $v = none ]; then spec=""; else spec=",$v"; fi
  (E2K_LANE_CUS="$spec" timeout 400 python bench.py $F --steps 20 --warmup 5) > gpurun_out/r06p_cfg3_${v/:/_}_${i}.log 2>&1
  python - ${v/:/_} $i <<'P'
import json, sys
for l in open(f'gpurun_out/r06p_cfg3_{sys.argv[1]}_{sys.argv[2]}.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print('cfg3 wgrad lane CUs', sys.argv[1], 'run', sys.argv[2], round(d['ms_per_step'], 3), 'launches', d['launches_per_step'])
P
done; done
