# Round 3, GPU call 14: first hardware run of the default-off switches (has_freq_axis, attn_laser, attn_fourier_embed_input), the
# K15 / K16 / K17 kernels and the plan replay with all switches on; then one short bench line (the default path must not have moved)
tag=${1:-r03n}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests/test_backbone.py tests/test_kernels_attn.py tests/test_e2tts.py tests/test_abi.py -m gpu -q \
  -k "fourier or laser or freq or cfg_combine or default_off or reference_golden or character_embed or duration or contract or test_backbone" \
  ) > gpurun_out/pytest_${tag}.log 2>&1
echo "[pytest] rc=$? $(tail -n 1 gpurun_out/pytest_${tag}.log)"
grep -E "FAILED|ERROR" gpurun_out/pytest_${tag}.log | head -20
(timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor) > gpurun_out/bench_${tag}.log 2>&1
echo "[bench] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}.log)"
