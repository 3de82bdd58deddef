# round 6: rotary q / k inside the attention projection's epilogue (e2k_gemm_nt_qkrot_bf16) on hardware -- tests, then sample() at cfg5 and
# the training step with E2K_FUSE_QK_ROT 0 / 1, interleaved on one box
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 1200 python -m pytest tests/test_kernels_gemm.py tests/test_backbone.py tests/test_kernels_attn.py tests/test_e2tts.py -m gpu -q -p no:cacheprovider -x -k "rotary or attention or golden or plan or sample or training") 2>&1 | tail -4
F="--no-cpu-baseline --no-optimizer-leg --no-launch-floor --no-warm-leg"
for i in 1 2 3; do
for v in 0 1; do
  (E2K_FUSE_QK_ROT=$v timeout 400 python bench.py $F --steps 20 --warmup 5) > gpurun_out/r06q_cfg3_${v}_${i}.log 2>&1
  python - $v $i <<'P'
import json, sys
for l in open(f'gpurun_out/r06q_cfg3_{sys.argv[1]}_{sys.argv[2]}.log'):
    if l.startswith('{'):
        d = json.loads(l); g = d['kernel_groups_ms_per_step']
        print('cfg3 E2K_FUSE_QK_ROT=' + sys.argv[1], 'run', sys.argv[2], round(d['ms_per_step'], 3), 'launches', d['launches_per_step'], 'qkv_post_fwd', g.get('qkv_post_fwd'), 'qkrot', g.get('gemm_nt_qkrot_bf16'), 'nt', g.get('gemm_nt_bf16'), 'frac', d['roofline']['frac'])
P
done; done
for i in 1 2; do
for v in 0 1; do
  (E2K_FUSE_QK_ROT=$v timeout 300 python tools/bench_sample.py) > gpurun_out/r06q_sample_${v}_${i}.log 2>&1
  echo "sample E2K_FUSE_QK_ROT=$v run $i: $(grep -o '"seconds": [0-9.]*' gpurun_out/r06q_sample_${v}_${i}.log) $(grep -o '"model_tflops_per_s": [0-9.]*' gpurun_out/r06q_sample_${v}_${i}.log)"
done; done
