# round 6: test_launch_lanes_match_single_stream fails about one run in three on the final tree (plans + lanes: the output of a training step
# differs from the single-stream schedule's) -- with which of the round's switches
export PYTHONUNBUFFERED=1
for i in 1 2 3 4 5 6; do
for v in "E2K_NOP=1" "HIP_FORCE_DEV_KERNARG=0" "E2K_FUSE_HC_NORM=0" "E2K_WGRAD_SPLITS_DUAL=0"; do
  r=$(env $v timeout 300 python -m pytest tests/test_backbone.py -m gpu -q -p no:cacheprovider -x -k "launch_lanes_match_single_stream" 2>&1 | grep -E "AssertionError|passed|failed" | tr '\n' ' ' | cut -c1-120)
  echo "$v run $i: $r"
done; done
