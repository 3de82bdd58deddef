# round 6, fourth call: is the graph replay's loss the cross-lane structure?  single-lane plans (E2K_LANES=0) eager vs graph, cfg2 + cfg3;
# trained_like parity case; resample / data path / cfg5 4-interval tests on hardware
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests/test_e2tts.py -m gpu -q -p no:cacheprovider -x -k "resample or foreign or hl_gauss") > gpurun_out/r06d_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/r06d_pytest.log
(timeout 900 python -m pytest tests/test_fullsize.py -m gpu -q -p no:cacheprovider -k "trained_like or cfg5_exact") > gpurun_out/r06d_pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -n 6 gpurun_out/r06d_pytest2.log
run() { t=$1; shift; (timeout 500 "$@") > gpurun_out/r06d_bench_$t.log 2>&1; echo "[$t] rc=$?"; python - "$t" <<'P'
import json, sys
for l in open(f'gpurun_out/r06d_bench_{sys.argv[1]}.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: (round(d[k], 3) if isinstance(d.get(k), float) else d.get(k)) for k in ('ms_per_step', 'host_enqueue_ms_per_step', 'mfma_roofline_frac_whole_step')})
P
}
F="--no-cpu-baseline --no-optimizer-leg --no-launch-floor --no-warm-leg"
run cfg2_1lane_eager env E2K_LANES=0 python bench.py --config cfg2 --steps 40 --warmup 10 --graphs 0 $F
run cfg2_1lane_graph env E2K_LANES=0 python bench.py --config cfg2 --steps 40 --warmup 10 --graphs 1 $F
run cfg3_1lane_eager env E2K_LANES=0 python bench.py --steps 20 --warmup 5 --graphs 0 $F
run cfg3_1lane_graph env E2K_LANES=0 python bench.py --steps 20 --warmup 5 --graphs 1 $F
