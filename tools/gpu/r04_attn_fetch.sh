# measurement only: FETCH_SIZE of the attention ring kernels with the plain and the XCD-aware workgroup numbering; MFMA busy of the weight-gradient GEMM
#   gpurun --timeout 300 -- 'bash tools/gpu/r04_attn_fetch.sh'
export PYTHONUNBUFFERED=1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
  (E2K_ATTN_XCD=$x timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc5_$x -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py attn) > $GRAFT_REPO_ROOT/gpurun_out/pmc5_xcd$x.log 2>&1; echo "fetch xcd=$x rc=$?"
  find /tmp/pmc5_$x -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/pmc5_xcd${x}_counters.csv \;
  find /tmp/pmc5_$x -name "*kernel_trace.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/pmc5_xcd${x}_trace.csv \;
done
(timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc6 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py tn) > $GRAFT_REPO_ROOT/gpurun_out/pmc6_tn.log 2>&1; echo "tn rc=$?"
find /tmp/pmc6 -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/pmc6_tn_counters.csv \;
