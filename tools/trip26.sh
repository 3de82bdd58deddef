mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 200 python tools/nt_shapes.py) 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
(timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d /tmp/pmc_tr -o p -- python $GRAFT_REPO_ROOT/tools/nt_traffic_probe.py) > $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic.log 2>&1; echo "pmc rc=$?"
find /tmp/pmc_tr -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/nt_traffic_counters.csv \;
ls -la $GRAFT_REPO_ROOT/gpurun_out/nt_traffic_counters.csv
