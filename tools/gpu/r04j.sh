# round 4, call j: SAME-BOX A/B of the attention kernel versions of this round (tools/ab/build_variants.py), two passes each
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04j
mkdir -p $O
cd $GRAFT_REPO_ROOT
for pass in 1 2; do
for v in r03 hash micro now setprio; do
  lib=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_$v.so; [ $v = now ] && lib=$GRAFT_REPO_ROOT/e2-tts-pytorch_amd/e2_tts_pytorch_amd/libe2k.so
  (E2K_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor) > $O/bench_${v}_$pass.log 2>&1
  echo "$v pass$pass $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${v}_$pass.log | head -1) $(grep -o '"attn_fwd": {[^}]*}' $O/bench_${v}_$pass.log | head -1) $(grep -o '"attn_bwd": {[^}]*}' $O/bench_${v}_$pass.log | head -1)" | tee -a $O/ab.txt
done
done
