# round 6: values or instruction form?  formBscalar: the rotary pair with the roundings of the failing form, b = fma(x1, c, x0 s), but compiled
# without packed instructions (the risky operand form cannot appear); formB: the failing form; default: today's
#   (build the variants first: python tools/ab/rotary_forms.py)
export PYTHONUNBUFFERED=1
for sd in 0 1 2; do
for v in formBscalar formB default; do
  lib=""; [ $v != default ] && lib="E2K_LIB=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_$v.so"
  echo "== $v seed $sd"; env HIP_FORCE_DEV_KERNARG=0 SEED=$sd $lib REPS=4 timeout 300 python tools/probes/lanes_race.py 2>&1 | grep -E "differs|gradient|done|Error" | cut -c1-150 | head -3
done; done
