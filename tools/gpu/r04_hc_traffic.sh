# measurement only: FETCH_SIZE / WRITE_SIZE of the hyper-connection kernels (separate passes)
export PYTHONUNBUFFERED=1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc7_$c -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py hc) > $GRAFT_REPO_ROOT/gpurun_out/pmc7_$c.log 2>&1; echo "hc $c rc=$?"
  find /tmp/pmc7_$c -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/pmc7_hc_$c.csv \;
done
