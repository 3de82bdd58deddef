# round 4, call k: same-box A/B of occupancy / vote variants (tools/ab/build_variants.py)
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04n
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in now halves now halves; do
  lib=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_$v.so; [ $v = now ] && lib=$GRAFT_REPO_ROOT/e2-tts-pytorch_amd/e2_tts_pytorch_amd/libe2k.so
  (E2K_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_$v.log 2>&1
  python - "$v" $O/bench_$v.log <<'PY' | tee -a $O/ab.txt
import sys, json
v, f = sys.argv[1], sys.argv[2]
j = json.loads([l for l in open(f) if l.startswith('{')][-1])
g = j['kernel_groups_ms_per_step']
pick = {k: g[k]['ms'] for k in ('attn_fwd', 'attn_bwd', 'gate_bwd', 'rmsnorm_bwd') if k in g}
print(v, round(j['ms_per_step'], 2), pick)
PY
done
