# round 5, call 2: attn32 with packed fp32 arithmetic -- A/B against ring16 (+ scalar-store publication, dQ at 2 / 3 waves per SIMD), tests, step
#   gpurun --timeout 900 -- 'bash tools/gpu/r05b.sh'
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05b
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 400 python3 -m pytest tests/test_kernels_attn.py -x -q -m gpu -p no:cacheprovider) > $O/pytest_attn.log 2>&1; echo "attn tests rc=$? $(tail -1 $O/pytest_attn.log)"
(timeout 300 python tools/probes/attn32_ab.py) > $O/attn32_ab.log 2>&1; echo "ab rc=$?"; tail -22 $O/attn32_ab.log
for v in 0 64 0 64; do
  (E2K_ATTN_FLAGS=$v timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_flags$v.log 2>&1; echo "[flags $v] rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_flags$v.log | head -1)"
done
