# Round 3, GPU call 10: the artefacts the bench line cites -- NT HBM traffic (PMC passes), single-stream kernel stats, the default bench line
# with cpu_baseline and launch floor, cfg2 line, sample() and MelSpec throughput, forced data-parallel A/B
tag=${1:-r03j}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash tools/gpu/traffic.sh
python tools/nt_traffic_reduce.py && cp profiles/r03_nt_traffic.json gpurun_out/r03_nt_traffic.json
(timeout 500 python bench.py --steps 20 --warmup 5 --dump-ops gpurun_out/r03_ops_by_shape.json) > gpurun_out/bench_${tag}_final.log 2>&1; echo "bench rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_final.log; grep -o '"roofline": {[^}]*}' gpurun_out/bench_${tag}_final.log | cut -c1-700
(timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor --force-ddp) > gpurun_out/bench_${tag}_ddp_bf16.log 2>&1; grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_ddp_bf16.log
(timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor --force-ddp --grad-dtype fp32) > gpurun_out/bench_${tag}_ddp_fp32.log 2>&1; grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_ddp_fp32.log
(timeout 300 python bench.py --config cfg2 --steps 20 --warmup 5 --no-cpu-baseline) > gpurun_out/bench_${tag}_cfg2.log 2>&1; grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_cfg2.log
bash tools/gpu/prof_single_stream.sh
