# round 4, call s: same-box sweep of the step-level switches that round 3 decided across leases (box-to-box spread 7 %), three
# baseline runs interleaved; sample() with and without the remainder split of the NT kernel
#   gpurun --timeout 1500 -- 'bash tools/gpu/r04s.sh'
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04s
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() {   # name, extra env
  local v=$1; shift
  (env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_$v.log 2>&1
  python - "$v" $O/bench_$v.log <<'PY' | tee -a $O/ab.txt
import sys, json
v, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(f) if l.startswith('{')][-1])
    g = j['kernel_groups_ms_per_step']
    pick = {k: g[k]['ms'] for k in ('gemm_nt_bf16', 'gemm_tn_group_bf16', 'gemm_tn_dual_bf16') if k in g}
    print(v, round(j['ms_per_step'], 2), j.get('launches_per_step'), pick)
except Exception as e:
    print(v, 'FAILED', e)
PY
}
run base1 X=0
run nosplit1 E2K_GEMM_FLAGS=16
run defer0_1 E2K_DEFER_REDUCES=0
run recast0_1 E2K_RECAST_T_ON_LANE=0
run wgroup0_1 E2K_WGRAD_GROUP=0
run t256min133_1 E2K_GEMM_T256_MIN=133
run base2 X=0
run t256min224_1 E2K_GEMM_T256_MIN=224
run wsplits2_1 E2K_WGRAD_SPLITS=2
run nosplit2 E2K_GEMM_FLAGS=16
run defer0_2 E2K_DEFER_REDUCES=0
run recast0_2 E2K_RECAST_T_ON_LANE=0
run wgroup0_2 E2K_WGRAD_GROUP=0
run base3 X=0
run nosplit_t133 E2K_GEMM_FLAGS=16 E2K_GEMM_T256_MIN=133
run batchred0 E2K_BATCH_REDUCES=0
run zerolane0 E2K_ZERO_GRADS_ON_LANE=0
run base4 X=0
(timeout 200 python tools/bench_sample.py) > $O/sample_default.log 2>&1; echo "sample default $(grep -o '"seconds": [0-9.]*' $O/sample_default.log)" | tee -a $O/ab.txt
(E2K_GEMM_FLAGS=16 timeout 200 python tools/bench_sample.py) > $O/sample_nosplit.log 2>&1; echo "sample nosplit $(grep -o '"seconds": [0-9.]*' $O/sample_nosplit.log)" | tee -a $O/ab.txt
