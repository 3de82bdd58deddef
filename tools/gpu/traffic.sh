# HBM-side traffic of the NT GEMM launch mix of a cfg3 step (profiles/r01_nt_traffic.json is computed from the CSVs):
#   gpurun --timeout 300 -- 'bash tools/gpu/traffic.sh'       (shape list: tools/nt_shapes_cfg3.json from tools/nt_shapes.py)
# FETCH_SIZE and WRITE_SIZE need separate passes (together: "exceeds the capabilities of the hardware to collect").
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/nt_traffic_probe.py) > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  find /tmp/pmc_$c -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/nt_traffic_$c.csv \;
done
