# round 4, call t: attention ring kernels with XCD-aware workgroup numbering (a (batch, head) row's tiles on one XCD), same-box A/B
#   gpurun --timeout 1500 -- 'bash tools/gpu/r04t.sh'
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04t
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 600 python3 -m pytest tests/test_kernels_attn.py -x -q -m gpu -p no:cacheprovider) > $O/pytest_attn.log 2>&1
echo "pytest rc=$? $(tail -1 $O/pytest_attn.log)"
run() {   # name, extra env
  local v=$1; shift
  (env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_$v.log 2>&1
  python - "$v" $O/bench_$v.log <<'PY' | tee -a $O/ab.txt
import sys, json
v, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(f) if l.startswith('{')][-1])
    g = j['kernel_groups_ms_per_step']
    pick = {k: g[k]['ms'] for k in ('attn_fwd', 'attn_bwd', 'gemm_nt_bf16') if k in g}
    print(v, round(j['ms_per_step'], 2), pick)
except Exception as e:
    print(v, 'FAILED', e)
PY
}
for i in 1 2 3; do
  run xcd1_$i E2K_ATTN_XCD=1
  run xcd0_$i E2K_ATTN_XCD=0
done
