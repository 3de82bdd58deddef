# round 4, call g: attention kernels after the MFMA reordering / writelane ballots / squeezed mask bits (tests + times), dK,dV at 4 waves per SIMD
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04h
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_attn.py tests/test_backbone.py -x -q -m gpu -p no:cacheprovider -k "not golden and not widths") > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for v in 0 0; do
(E2K_ATTN_FLAGS=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_$v.log 2>&1
echo "attn_flags=$v $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.log | head -1) $(grep -o '"attn_fwd": {[^}]*}' $O/bench_$v.log | head -1) $(grep -o '"attn_bwd": {[^}]*}' $O/bench_$v.log | head -1)" | tee -a $O/ab.txt
done
