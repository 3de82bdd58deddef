# round 4, call f: attention with the full-rate dropout hash and the squeezed key-mask bits (tests + per-kernel times)
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_attn.py tests/test_backbone.py -x -q -m gpu -p no:cacheprovider -k "not golden and not widths") > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for i in 1 2; do
(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench$i.log 2>&1
echo "bench$i $(grep -o '"ms_per_step": [0-9.]*' $O/bench$i.log | head -1) $(grep -o '"attn_fwd": {[^}]*}' $O/bench$i.log | head -1) $(grep -o '"attn_bwd": {[^}]*}' $O/bench$i.log | head -1)"
done
