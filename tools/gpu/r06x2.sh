# round 6: tight-loop two-kernel reproducer (tools/probes/pk_neg_broadcast2.py); variants: python tools/ab/rotary_forms.py
#   E2K_GEMM_FLAGS 256 = the 128 x 128 LDS-DMA GEMM for every shape, 257 = the same GEMMs staging through registers (no LDS-DMA)
export PYTHONUNBUFFERED=1
for v in formB formBscalar oldrot default; do for gf in 256 257; do
  lib=""; [ $v != default ] && lib="E2K_LIB=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_$v.so"
  echo "== $v E2K_GEMM_FLAGS=$gf"; env $lib E2K_GEMM_FLAGS=$gf timeout 300 python tools/probes/pk_neg_broadcast2.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-600
done; done
