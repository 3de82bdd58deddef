# round 6, first call: the whole GPU suite on the tree with the tightened parity checks + the bench line with the warm leg
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10) > gpurun_out/r06a_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 30 gpurun_out/r06a_pytest.log
(timeout 600 python bench.py --steps 20 --warmup 5) > gpurun_out/r06a_bench.log 2>&1; echo "bench rc=$?"
python - <<'P'
import json
for l in open('gpurun_out/r06a_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print({k:d.get(k) for k in ('ms_per_step','ms_per_step_warm','launches_per_step','host_enqueue_ms_per_step')}, d['roofline']['frac'])
P
