# round 6: does the round-3 fence (tests/test_concurrency.py: every kernel family next to both LDS-DMA GEMM kinds, 200 trials) catch the
# rotary form that failed under the launch lanes?  (variant libraries: python tools/ab/rotary_forms.py)
export PYTHONUNBUFFERED=1
for v in formB formBscalar default; do
  lib=""; [ $v != default ] && lib="E2K_LIB=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_$v.so"
  echo "== $v"; env $lib timeout 600 python -m pytest tests/test_concurrency.py -m gpu -q -p no:cacheprovider -k "every_kernel or concurrent_gemm" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-400 | head -6
done
