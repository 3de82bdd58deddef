# round 5: optimizer-side kernels alone, variants (tools/probes/optim_ab.py), then the optimizer tests on hardware
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05k
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python tools/probes/optim_ab.py $O/optim_ab.json) 2>&1 | tee $O/optim_ab.log | cut -c1-400
(timeout 600 python -m pytest tests/test_optim.py -x -q -m gpu -p no:cacheprovider) > $O/pytest_optim.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_optim.log)"
