# round 5: cfg2 -- which NT kernel for its 66- / 198-tile outputs (E2K_GEMM_T256_MIN), same box
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05o
mkdir -p $O
cd $GRAFT_REPO_ROOT
for t in 64 67 199 64 67 199; do
  (E2K_GEMM_T256_MIN=$t timeout 300 python bench.py --config cfg2 --steps 30 --warmup 5 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/cfg2_t$t.log 2>&1
  echo "t256_min=$t $(grep -o '"ms_per_step": [0-9.]*' $O/cfg2_t$t.log) $(grep -o '"gemm_nt_bf16": {[^}]*}' $O/cfg2_t$t.log)" | tee -a $O/ab.txt
done
