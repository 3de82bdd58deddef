# round 5: pre-check of the V-less no-grad forward: sample / golden tests, sample() at cfg5
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05s
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_e2tts.py tests/test_backbone.py -x -q -m gpu -p no:cacheprovider) > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
(timeout 300 python tools/bench_sample.py) > $O/sample.log 2>&1; echo "$(tail -1 $O/sample.log | cut -c1-220)"
