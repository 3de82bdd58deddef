# round 5: NT epilogue store drain -- per-CU cost or the chip's HBM write rate (tools/probes/nt_store_drain.py)
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05q
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 600 python tools/probes/nt_store_drain.py $O/nt_store_drain.json) 2>&1 | tee $O/log.txt | grep -v Warn
