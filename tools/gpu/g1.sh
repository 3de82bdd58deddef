mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
(timeout 600 python -m pytest tests/test_kernels_attn.py tests/test_kernels_gemm.py tests/test_backbone.py -m gpu -q -p no:cacheprovider -x) 2>&1 | tail -5
(timeout 200 python tools/microbench.py attn) 2>&1 | grep attn_
(timeout 200 env E2K_ATTN_FLAGS=128 python tools/microbench.py attn) 2>&1 | grep attn_
(timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline) > gpurun_out/bench_g1.log 2>&1; python - <<'PY'
import json
for l in open('gpurun_out/bench_g1.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step']); print({k:(v['ms'],v.get('tflops')) for k,v in list(d['kernel_groups_ms_per_step'].items())[:12]})
PY
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mb -- python $GRAFT_REPO_ROOT/tools/microbench.py attn) > /dev/null 2>&1
find /tmp/prof_mb -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_mb_attn_kernel_stats.csv \;
grep -i attn $GRAFT_REPO_ROOT/gpurun_out/prof_mb_attn_kernel_stats.csv | cut -c1-160
