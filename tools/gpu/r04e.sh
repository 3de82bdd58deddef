# round 4, call e: adaptive tile-group size of the 256 x 256 NT kernel (time + HBM-side traffic), sample() at cfg5
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04e
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_kernels_gemm.py -x -q -m gpu -p no:cacheprovider) > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc=$? $(tail -1 $O/pytest_gemm.log)"
for v in 0 8 0 8; do
  (E2K_GEMM_GROUP=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_group$v.log 2>&1
  echo "gemm_group=$v $(grep -o '"ms_per_step": [0-9.]*' $O/bench_group$v.log | head -1) nt=$(grep -o '"gemm_nt_bf16": {"ms": [0-9.]*' $O/bench_group$v.log | head -1)" | tee -a $O/group_ab.txt
done
(timeout 400 python tools/bench_sample.py) > $O/sample.log 2>&1; tail -2 $O/sample.log
cd /tmp && export TMPDIR=/tmp
for v in 0 8; do
for c in FETCH_SIZE WRITE_SIZE; do
  (E2K_GEMM_GROUP=$v timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${c}_$v -o p -- python $GRAFT_REPO_ROOT/tools/nt_traffic_probe.py) > $O/pmc_${c}_$v.log 2>&1; echo "pmc $c group=$v rc=$?"
  find /tmp/pmc_${c}_$v -name "*counter_collection.csv" -exec cp {} $O/nt_traffic_${c}_group$v.csv \;
done
done
