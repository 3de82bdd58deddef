# round 5, call 9: fused GEGLU backward on hardware (tests + step A/B)
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05i
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 400 python3 -m pytest tests/test_kernels_gemm.py -x -q -m gpu -p no:cacheprovider -k "geglu") > $O/pytest_geglu.log 2>&1; echo "geglu tests rc=$? $(tail -1 $O/pytest_geglu.log)"
(timeout 400 python3 -m pytest tests/test_backbone.py -x -q -m gpu -p no:cacheprovider -k "test_backbone") > $O/pytest_backbone.log 2>&1; echo "backbone tests rc=$? $(tail -1 $O/pytest_backbone.log)"
for v in 1 0 1 0; do
  (E2K_FUSE_GEGLU_BWD=$v timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_fuse$v.log 2>&1; echo "[fuse_geglu_bwd $v] rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_fuse$v.log | head -1)"
done
