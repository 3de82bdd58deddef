# round 4, call p: same-box A/B of the row-kernel package (depthwise-conv kernels with tap windows, 8-wave rmsnorm_bwd / gate_bwd,
# conv_reduce with 8 loads in flight) and of the remainder-split policy of the 256 x 256 NT kernel
#   gpurun --timeout 1500 -- 'bash tools/gpu/r04p.sh'
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04p
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 600 python3 -m pytest tests/test_kernels_elementwise.py tests/test_kernels_gemm.py -x -q -m gpu -p no:cacheprovider) > $O/pytest_rows.log 2>&1
echo "pytest rc=$? $(tail -1 $O/pytest_rows.log)"
run() {   # name, lib, extra env
  local v=$1 lib=$2; shift 2
  (env E2K_LIB=$lib "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_$v.log 2>&1
  python - "$v" $O/bench_$v.log <<'PY' | tee -a $O/ab.txt
import sys, json
v, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(f) if l.startswith('{')][-1])
    g = j['kernel_groups_ms_per_step']
    pick = {k: g[k]['ms'] for k in ('dwconv_fwd', 'dwconv_bwd', 'dwconv_bwd_reduce', 'gate_bwd', 'rmsnorm_bwd', 'gemm_nt_bf16') if k in g}
    print(v, round(j['ms_per_step'], 2), pick)
except Exception as e:
    print(v, 'FAILED', e)
PY
}
NOW=$GRAFT_REPO_ROOT/e2-tts-pytorch_amd/e2_tts_pytorch_amd/libe2k.so
L=$GRAFT_REPO_ROOT/tools/ab/lib
run base $L/libe2k_base.so
run now $NOW
run convonly $L/libe2k_convonly.so
run rms128 $L/libe2k_rms128.so
run conv768 $L/libe2k_conv768.so
run nosplit $NOW E2K_GEMM_FLAGS=16
run splitcap2 $NOW E2K_GEMM_SPLIT_CAP=2
run splitmink2 $NOW E2K_GEMM_SPLIT_MINK=2
run base2 $L/libe2k_base.so
run now2 $NOW
