# round 5: hyper-connection kernels with non-temporal row loads / stores (variant libraries), same box
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05u
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in default hc_ntld hc_ntst hc_ntboth default hc_ntboth; do
  L=""; [ $v != default ] && L=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_$v.so
  (E2K_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/bench_$v.log 2>&1
  echo "$v $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.log | head -1) $(grep -o '"hc_bwd": {[^}]*}' $O/bench_$v.log) $(grep -o '"hc_fwd": {[^}]*}' $O/bench_$v.log)"
done 2>&1 | tee $O/ab.txt
