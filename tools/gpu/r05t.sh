# round 5: NT traffic with the remainder workgroups first (default) against last (variant library), same box: FETCH_SIZE over the step's NT mix,
# and the step time
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in first last first last; do
  L=""; [ $v = last ] && L=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_remlast.so
  rm -rf /tmp/pmc_$v
  (E2K_LIB=$L timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_$v -o p -- python $GRAFT_REPO_ROOT/tools/nt_traffic_probe.py) > $O/pmc_$v.log 2>&1
  f=$(find /tmp/pmc_$v -name "*counter_collection.csv" | head -1)
  python3 - "$f" $v <<'PY'
import csv, sys
tot = 0.; n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == 'FETCH_SIZE' and 'gemm_nt' in r['Kernel_Name']:
        tot += float(r['Counter_Value']); n += 1
print('rem', sys.argv[2], 'FETCH_SIZE sum over the probe (KB units x2 on gfx950):', tot, 'dispatches', n)
PY
done 2>&1 | tee $O/fetch_ab.txt
cd $GRAFT_REPO_ROOT
for v in first last first last; do
  L=""; [ $v = last ] && L=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_remlast.so
  (E2K_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/bench_$v.log 2>&1
  echo "rem $v $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.log | head -1) $(grep -o '"gemm_nt_bf16": {[^}]*}' $O/bench_$v.log)"
done 2>&1 | tee $O/step_ab.txt
