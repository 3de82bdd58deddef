# round 6, third call (graph capture fixed): HIP-graph replay of the recorded plans -- bit-identity tests on hardware, then same-box A/B of the eager replay
# against the graph replay at cfg3 and cfg2 (+ the host launch floor in both modes, inside each bench line), trained_like parity case
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests/test_backbone.py tests/test_e2tts.py tests/test_plan_lanes.py -m gpu -q -p no:cacheprovider -x -k "graph or plan or hl_gauss or duration") > gpurun_out/r06c_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 gpurun_out/r06c_pytest.log
(timeout 600 python -m pytest tests/test_fullsize.py -m gpu -q -p no:cacheprovider -k "trained_like") > gpurun_out/r06c_pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -n 12 gpurun_out/r06c_pytest2.log
run() { t=$1; shift; (timeout 500 "$@") > gpurun_out/r06c_bench_$t.log 2>&1; echo "[$t] rc=$?"; python - "$t" <<'P'
import json, sys
for l in open(f'gpurun_out/r06c_bench_{sys.argv[1]}.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: (round(d[k], 3) if isinstance(d.get(k), float) else d.get(k)) for k in ('ms_per_step', 'ms_per_step_warm', 'host_enqueue_ms_per_step', 'mfma_roofline_frac_whole_step')},
              'floor', {k: round(v, 2) for k, v in (d.get('host_launch_floor') or {}).items() if isinstance(v, float)},
              'floor_other', {k: round(v, 2) for k, v in (d.get('host_launch_floor_other_mode') or {}).items() if isinstance(v, float)})
P
}
run cfg3_eager_replay python bench.py --steps 20 --warmup 5 --graphs 0 --no-cpu-baseline --no-optimizer-leg
run cfg3_graph python bench.py --steps 20 --warmup 5 --graphs 1 --no-cpu-baseline --no-optimizer-leg
run cfg2_eager_replay python bench.py --config cfg2 --steps 40 --warmup 10 --graphs 0 --no-cpu-baseline --no-optimizer-leg --no-launch-floor
run cfg2_graph python bench.py --config cfg2 --steps 40 --warmup 10 --graphs 1 --no-cpu-baseline --no-optimizer-leg --no-launch-floor
run cfg3_eager_replay2 python bench.py --steps 20 --warmup 5 --graphs 0 --no-cpu-baseline --no-optimizer-leg --no-launch-floor --no-warm-leg
run cfg3_graph2 python bench.py --steps 20 --warmup 5 --graphs 1 --no-cpu-baseline --no-optimizer-leg --no-launch-floor --no-warm-leg
