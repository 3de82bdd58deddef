# measurement only: MFMA pipe busy cycles of the NT GEMM and attention kernels (rocprofv3 --pmc, kernel trace only)
#   gpurun --timeout 300 -- 'bash tools/gpu/r04_mfma_pmc.sh'
export PYTHONUNBUFFERED=1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in nt attn; do
  (timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d /tmp/pmc4_$w -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py $w) > $GRAFT_REPO_ROOT/gpurun_out/pmc4_$w.log 2>&1; echo "pmc4 $w rc=$?"
  find /tmp/pmc4_$w -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/pmc4_${w}_counters.csv \;
done
