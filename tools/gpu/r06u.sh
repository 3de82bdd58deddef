# round 6: rot_pair pinned to the contraction the compiler had chosen for the old expression (b = fma(x0, s, x1 c)): probe on four models,
# then the fused-projection tests and the lanes test
export PYTHONUNBUFFERED=1
for sd in 0 1 2 3; do
for ka in 0 1; do
  echo "== seed $sd HIP_FORCE_DEV_KERNARG=$ka"; env HIP_FORCE_DEV_KERNARG=$ka SEED=$sd REPS=4 timeout 300 python tools/probes/lanes_race.py 2>&1 | grep -E "differs|gradient|done|Error" | cut -c1-170 | head -4
done; done
(timeout 900 python -m pytest tests/test_kernels_gemm.py tests/test_backbone.py -m gpu -q -p no:cacheprovider -k "rotary or lanes or plan") 2>&1 | tail -4
