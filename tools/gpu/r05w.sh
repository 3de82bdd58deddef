# round 5: depthwise conv with 16-byte output stores through LDS against 4-byte stores from the registers (variant library = HEAD's elementwise.hip)
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05w
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_elementwise.py tests/test_backbone.py -x -q -m gpu -p no:cacheprovider) > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for v in new old new old; do
  L=""; [ $v = old ] && L=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_ew_old.so
  (E2K_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/bench_$v.log 2>&1
  echo "$v $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.log | head -1) $(grep -o '"dwconv_bwd": {[^}]*}' $O/bench_$v.log) $(grep -o '"dwconv_fwd": {[^}]*}' $O/bench_$v.log)"
done 2>&1 | tee $O/ab.txt
for v in new old; do
  L=""; [ $v = old ] && L=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_ew_old.so
  (E2K_LIB=$L timeout 300 python tools/bench_sample.py 32 8) > $O/sample_$v.log 2>&1; echo "sample(8 steps) $v $(tail -1 $O/sample_$v.log | grep -o '"seconds": [0-9.]*')"
done 2>&1 | tee -a $O/ab.txt
