# round 5: optimizer kernels with the larger update grid (variants), the bench's interleaved optimizer leg
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05m
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python tools/probes/optim_ab.py $O/optim_ab.json) 2>&1 | tee $O/optim_ab.log | cut -c1-330
(timeout 600 python bench.py --no-cpu-baseline --no-launch-floor) > $O/bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r05m/bench.log'):
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], json.dumps({k: v for k, v in d['optimizer_leg'].items() if k != 'note'}))
PY
(timeout 600 python -m pytest tests/test_optim.py -x -q -m gpu -p no:cacheprovider) > $O/pytest_optim.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_optim.log)"
