# round 5, call 1: the new attention kernels on hardware -- correctness (attention tests, new parity tests), same-box A/B against the
# first-generation ring kernels, scalar-store publication probe, step A/B
#   gpurun --timeout 1500 -- 'bash tools/gpu/r05a.sh'
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05a
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 400 python3 -m pytest tests/test_kernels_attn.py -x -q -m gpu -p no:cacheprovider) > $O/pytest_attn.log 2>&1; echo "attn tests rc=$? $(tail -1 $O/pytest_attn.log)"
(E2K_ATTN_FLAGS=64 timeout 300 python3 -m pytest tests/test_kernels_attn.py -x -q -m gpu -p no:cacheprovider -k "dropout") > $O/pytest_attn_ring16.log 2>&1; echo "attn dropout tests, ring16 rc=$? $(tail -1 $O/pytest_attn_ring16.log)"
(E2K_ATTN32_PUB=1 timeout 300 python3 -m pytest tests/test_kernels_attn.py -x -q -m gpu -p no:cacheprovider -k "dropout") > $O/pytest_attn_sstore.log 2>&1; echo "attn dropout tests, scalar stores rc=$? $(tail -1 $O/pytest_attn_sstore.log)"
(timeout 300 python tools/probes/attn32_ab.py) > $O/attn32_ab.log 2>&1; echo "ab rc=$?"; tail -25 $O/attn32_ab.log
for v in 0 64 0 64; do
  (E2K_ATTN_FLAGS=$v timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_flags$v.log 2>&1; echo "[flags $v] rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_flags$v.log | head -1)"
done
(timeout 900 python3 -m pytest tests/test_e2tts.py tests/test_fullsize.py -x -q -m gpu -p no:cacheprovider -k "dropout_against or fed_masks or batch8_forward_backward or half_randomized") > $O/pytest_parity.log 2>&1; echo "parity tests rc=$? $(tail -1 $O/pytest_parity.log)"
grep -h "cfg3\|loss rel" $O/pytest_parity.log | head
