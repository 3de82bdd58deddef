# round 4, last call: measurement only (no tree change): cfg2 bench line, forced data-parallel exchange at world size 1 (fp32 / bf16 slabs),
# rocprofv3 kernel stats of the default (launch lanes on) bench
#   gpurun --timeout 420 -- 'bash tools/gpu/r04_extras.sh'
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_extras
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 120 python bench.py --config cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/bench_cfg2.log 2>&1; echo "cfg2 rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_cfg2.log | head -1)"
(timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/bench_plain.log 2>&1; echo "plain rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_plain.log | head -1)"
(MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 120 python bench.py --force-ddp --grad-dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/bench_ddp_fp32.log 2>&1; echo "ddp fp32 rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ddp_fp32.log | head -1)"
(MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 timeout 120 python bench.py --force-ddp --grad-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/bench_ddp_bf16.log 2>&1; echo "ddp bf16 rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ddp_bf16.log | head -1)"
cd /tmp && export TMPDIR=/tmp
(timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lanes -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/prof_lanes.log 2>&1; echo "prof rc=$?"
find /tmp/prof_lanes -name "*kernel_stats.csv" -exec cp {} $O/prof_lanes_kernel_stats.csv \;
