# Round 3, GPU call 15: refresh of the judged artefacts on the final tree -- the bench line exactly as the driver runs it (default
# flags: launch floor, optimizer leg, CPU baseline), the single-stream rocprofv3 kernel summary, cfg5 sample(), the whole -m gpu suite
tag=${1:-r03o}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 600 python bench.py --steps 20 --warmup 5) > gpurun_out/bench_${tag}_default.log 2>&1
echo "[bench default] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_default.log) $(grep -o '"clip_adopt_ms": [0-9.]*' gpurun_out/bench_${tag}_default.log) $(grep -o '"frac": [0-9.]*' gpurun_out/bench_${tag}_default.log | head -1)"
bash tools/gpu/prof_single_stream.sh
(timeout 300 python tools/bench_sample.py) > gpurun_out/sample_${tag}.log 2>&1; echo "[sample] rc=$? $(tail -n 1 gpurun_out/sample_${tag}.log | cut -c1-200)"
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider) > gpurun_out/pytest_${tag}.log 2>&1; echo "[pytest] rc=$?"; tail -n 8 gpurun_out/pytest_${tag}.log
