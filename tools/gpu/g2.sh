mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests/test_backbone.py tests/test_e2tts.py tests/test_fullsize.py -m gpu -q -p no:cacheprovider -x) 2>&1 | tail -5
for l in 1 0; do E2K_LANES=$l timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | grep -o "\"ms_per_step\": [0-9.]*\|\"loss\": [0-9.a-z]*\|\"host_enqueue_ms_per_step\": [0-9.]*" | tr "\n" " "; echo " plan lanes=$l"; done
for l in 1 0; do E2K_LANES=$l timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --eager 2>&1 | grep -o "\"ms_per_step\": [0-9.]*\|\"loss\": [0-9.a-z]*" | tr "\n" " "; echo " eager lanes=$l"; done
