# round 4, call u (measurement only, variant library, nothing landed): the K-split remainder workgroups of the 256 x 256 NT kernel FIRST
# in the grid (tools/ab/patches/gemm_rem_first.patch) against the product library, one box
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04u
mkdir -p $O
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_remfirst.so
(E2K_LIB=$L timeout 200 python3 -m pytest tests/test_kernels_gemm.py -x -q -m gpu -p no:cacheprovider) > $O/pytest_gemm.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_gemm.log)"
run() {
  local v=$1; shift
  (env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_$v.log 2>&1
  python - "$v" $O/bench_$v.log <<'PY' | tee -a $O/ab.txt
import sys, json
v, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(v, round(j['ms_per_step'], 2), j['kernel_groups_ms_per_step']['gemm_nt_bf16'])
except Exception as e:
    print(v, 'FAILED', e)
PY
}
for i in 1 2 3; do
  run now$i X=0
  run remfirst$i E2K_LIB=$L
done
