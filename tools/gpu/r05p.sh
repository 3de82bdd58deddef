# round 5: sample() with the conditional / null passes on two streams against one after the other (same box), the bit-identity test
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05p
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_e2tts.py -x -q -m gpu -k "concurrent or sample" -p no:cacheprovider) > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for c in 1 0 1 0; do
  (E2K_CFG_CONCURRENT=$c timeout 300 python tools/bench_sample.py) > $O/sample_c$c.log 2>&1; echo "concurrent=$c $(tail -1 $O/sample_c$c.log | cut -c1-200)"
done
