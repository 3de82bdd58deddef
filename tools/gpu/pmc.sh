# PMC counter passes over one kernel family:  gpurun --timeout 600 -- 'bash tools/gpu/pmc.sh nt|tn|attn|hc'
#   (counters in separate rocprofv3 runs, kernel-trace only: gpurun refuses --pmc together with sys/hip/hsa traces)
w=${1:-nt}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
(timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$w -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py $w) > $GRAFT_REPO_ROOT/gpurun_out/pmc_$w.log 2>&1; echo "pmc $w rc=$?"
(timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d /tmp/pmc2_$w -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py $w) > $GRAFT_REPO_ROOT/gpurun_out/pmc2_$w.log 2>&1; echo "pmc2 $w rc=$?"
(timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d /tmp/pmc3_$w -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py $w) > $GRAFT_REPO_ROOT/gpurun_out/pmc3_$w.log 2>&1; echo "pmc3 $w rc=$?"
for k in pmc pmc2 pmc3; do find /tmp/${k}_$w -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${k}_${w}_counters.csv \; ; done
