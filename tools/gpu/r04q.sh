# round 4, call q: same-box A/B: attention forward compiled for five waves per SIMD (key mask by ballot, 32 KB of LDS), the row-kernel
# package with 768 conv workgroups + single-wave conv_reduce, and the step with / without the remainder split of the NT kernel
#   gpurun --timeout 1500 -- 'bash tools/gpu/r04q.sh'
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04q
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 600 python3 -m pytest tests/test_kernels_elementwise.py tests/test_kernels_attn.py -x -q -m gpu -p no:cacheprovider) > $O/pytest_kernels.log 2>&1
echo "pytest rc=$? $(tail -1 $O/pytest_kernels.log)"
run() {   # name, lib, extra env
  local v=$1 lib=$2; shift 2
  (env E2K_LIB=$lib "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_$v.log 2>&1
  python - "$v" $O/bench_$v.log <<'PY' | tee -a $O/ab.txt
import sys, json
v, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(f) if l.startswith('{')][-1])
    g = j['kernel_groups_ms_per_step']
    pick = {k: g[k]['ms'] for k in ('attn_fwd', 'attn_bwd', 'dwconv_fwd', 'dwconv_bwd', 'dwconv_bwd_reduce', 'gate_bwd', 'rmsnorm_bwd', 'gemm_nt_bf16') if k in g}
    print(v, round(j['ms_per_step'], 2), pick)
except Exception as e:
    print(v, 'FAILED', e)
PY
}
NOW=$GRAFT_REPO_ROOT/e2-tts-pytorch_amd/e2_tts_pytorch_amd/libe2k.so
L=$GRAFT_REPO_ROOT/tools/ab/lib
for i in 1 2 3; do
  run now$i $NOW
  run attn_old$i $L/libe2k_attn_old.so
  run elem_old$i $L/libe2k_elem_old.so
  run nosplit$i $NOW E2K_GEMM_FLAGS=16
done
