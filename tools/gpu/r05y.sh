# round 5: the data-parallel wrapper on one GPU (world size 1 through RCCL): the path the driver's N > 1 bench takes, with both wire formats
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05y
mkdir -p $O
cd $GRAFT_REPO_ROOT
for g in bf16 fp32; do
  (timeout 300 python bench.py --force-ddp --grad-dtype $g --steps 10 --warmup 3 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/bench_ddp_$g.log 2>&1
  echo "force-ddp $g rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ddp_$g.log | head -1) $(grep -o '"loss": [0-9.]*' $O/bench_ddp_$g.log | head -1)"
done
tail -3 $O/bench_ddp_bf16.log | cut -c1-300
