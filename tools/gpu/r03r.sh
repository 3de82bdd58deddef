# Round 3, GPU call 18: dropout keep bits ahead of the forward (fourth launch lane) -- test on hardware, then A/B of the step.
# (Record of a measurement: the generator kernel, the forward's read-the-bits mode and E2K_DROPBITS_AHEAD were removed after it --
#  step-neutral, profiles/r03_dropbits_ahead_ab.jsonl; the code is in the history at the commit named in profiles/README.md.)
tag=${1:-r03r}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 600 python -m pytest tests/test_backbone.py tests/test_kernels_attn.py -m gpu -q -k "keep_bits_ahead or shared_dropout or other_widths") > gpurun_out/pytest_${tag}.log 2>&1
echo "[pytest] rc=$? $(tail -n 1 gpurun_out/pytest_${tag}.log)"
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log)"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-launch-floor --no-optimizer-leg"
run base $B
run ahead env E2K_DROPBITS_AHEAD=1 $B
run base2 $B
run ahead2 env E2K_DROPBITS_AHEAD=1 $B
