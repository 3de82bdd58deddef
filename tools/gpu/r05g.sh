# round 5, call 7: mixed backward (dQ 32 rows per wave + first-generation dK/dV on the new mask layout) A/B, attention tests, step A/B
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05g
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 400 python3 -m pytest tests/test_kernels_attn.py -x -q -m gpu -p no:cacheprovider) > $O/pytest_attn.log 2>&1; echo "attn tests rc=$? $(tail -1 $O/pytest_attn.log)"
(timeout 300 python tools/probes/attn32_ab.py) > $O/attn32_ab.log 2>&1; echo "ab rc=$?"; tail -18 $O/attn32_ab.log
for v in 0 64 0 64; do
  (E2K_ATTN_FLAGS=$v timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_flags$v.log 2>&1; echo "[flags $v] rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_flags$v.log | head -1)"
done
