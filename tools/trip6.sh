mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
for w in nt attn; do
  (timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pmc_$w -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py $w) > $GRAFT_REPO_ROOT/gpurun_out/pmc_$w.log 2>&1; echo "pmc $w rc=$?"
  (timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d /tmp/pmc2_$w -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py $w) > $GRAFT_REPO_ROOT/gpurun_out/pmc2_$w.log 2>&1; echo "pmc2 $w rc=$?"
  find /tmp/pmc_$w -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/pmc_${w}_counters.csv \;
  find /tmp/pmc2_$w -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/pmc2_${w}_counters.csv \;
done
ls -la $GRAFT_REPO_ROOT/gpurun_out/ | grep pmc
