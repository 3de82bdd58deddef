# round 6: the failing rotary form (variant library formB) under the launch lanes -- does it need LDS-DMA traffic next to it?  E2K_GEMM_FLAGS=1:
# the GEMMs stage their operands through registers (no global_load_lds); E2K_ATTN_FLAGS=128: the register-staged attention kernels
export PYTHONUNBUFFERED=1
L="E2K_LIB=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_formB.so"
for i in 1 2 3; do
for v in "E2K_NOP=1" "E2K_GEMM_FLAGS=1" "E2K_ATTN_FLAGS=128" "E2K_GEMM_FLAGS=1 E2K_ATTN_FLAGS=128"; do
  echo "== formB $v run $i"; env HIP_FORCE_DEV_KERNARG=0 $L $v SEED=$i REPS=4 timeout 300 python tools/probes/lanes_race.py 2>&1 | grep -E "differs|gradient|done|Error" | cut -c1-140 | head -2
done; done
