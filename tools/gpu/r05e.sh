# round 5, call 5: PMC passes over the attn32 kernels (what bounds them?)
#   gpurun --timeout 900 -- 'bash tools/gpu/r05e.sh'
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
bash tools/gpu/pmc.sh attn
cd /tmp && export TMPDIR=/tmp
(timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /tmp/pmc4_attn -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py attn) > $GRAFT_REPO_ROOT/gpurun_out/pmc4_attn.log 2>&1; echo "pmc4 rc=$?"
find /tmp/pmc4_attn -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/pmc4_attn_counters.csv \;
(timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_attn -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py attn) > $GRAFT_REPO_ROOT/gpurun_out/kt_attn.log 2>&1
find /tmp/kt_attn -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r05e_attn_kernel_stats.csv \;
head -12 $GRAFT_REPO_ROOT/gpurun_out/r05e_attn_kernel_stats.csv | cut -c1-200
