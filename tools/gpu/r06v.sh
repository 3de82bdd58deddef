# round 6: the post kernel alone against itself next to other kernels on another stream -- today's library and the variant with the rotary pair in
# the form that compiles to a one-lane negation of a broadcast packed-fp32 source; the three ways the backbone calls it
export PYTHONUNBUFFERED=1
for c in nograd later_layer; do
for v in default formB; do
  lib=""; [ $v != default ] && lib="E2K_LIB=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_$v.so"
  echo "== $v"; env $lib CFG=$c ITERS=400 timeout 300 python tools/probes/pk_neg_broadcast.py 2>&1 | grep -v amdgpu.ids | tail -3
done; done
