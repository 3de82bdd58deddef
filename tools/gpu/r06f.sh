# round 6: does the NT remainder split (157 fix-up launches per step, fp32 partial traffic) still pay IN the step, next to the launch lanes?
# same-box A/B: default vs E2K_GEMM_FLAGS=16 (never split), two runs each, interleaved
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
F="--no-cpu-baseline --no-optimizer-leg --no-launch-floor --no-warm-leg --steps 20 --warmup 5"
for i in 1 2; do
for v in default nosplit; do
  fl=0; [ $v = nosplit ] && fl=16
  (E2K_GEMM_FLAGS=$fl timeout 400 python bench.py $F) > gpurun_out/r06f_${v}_$i.log 2>&1
  python - $v $i <<'P'
import json, sys
for l in open(f'gpurun_out/r06f_{sys.argv[1]}_{sys.argv[2]}.log'):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[1], sys.argv[2], round(d['ms_per_step'], 2), 'launches', d['launches_per_step'], 'nt frac', round(d['roofline']['frac'], 4), 'nt ms', d['kernel_groups_ms_per_step']['gemm_nt_bf16'])
P
done; done
