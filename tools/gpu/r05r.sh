# round 5: cfg5-exact parity case (B 32, 1024 frames, cfg3 dims; oracle on three rows)
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05r
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_fullsize.py -x -q -m gpu -k "cfg5_exact" -s -p no:cacheprovider) > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"; grep "rel-L2" $O/pytest.log
