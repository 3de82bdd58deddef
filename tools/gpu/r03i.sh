# Round 3, GPU call 9: grouped weight-gradient launches (hardware tests + bench A/B), the new config-gap parity cases
tag=${1:-r03i}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 300 python -m pytest tests/test_kernels_gemm.py tests/test_backbone.py -m gpu -q -p no:cacheprovider -x -k 'tn or dual_source') > gpurun_out/pytest_${tag}_tn.log 2>&1; echo "pytest tn rc=$?"; tail -n 3 gpurun_out/pytest_${tag}_tn.log
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log) $(grep -o '"gemm_tn_bf16": {[^}]*}' gpurun_out/bench_${tag}_$t.log | head -1) $(grep -o '"gemm_tn_group_bf16": {[^}]*}' gpurun_out/bench_${tag}_$t.log | head -1) $(grep -o '"launches_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log | head -1)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | tail -n 2; }
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor"
run group $B
run no_group env E2K_WGRAD_GROUP=0 $B
run group_again $B
run group_splits4 env E2K_WGRAD_SPLITS=4 $B
(timeout 600 python -m pytest tests/test_fullsize.py -m gpu -q -p no:cacheprovider -k 'cfg2 or cfg3_dims_reference or at_cfg3' --durations=4) > gpurun_out/pytest_${tag}_fullsize.log 2>&1; echo "pytest fullsize rc=$?"; tail -n 8 gpurun_out/pytest_${tag}_fullsize.log
