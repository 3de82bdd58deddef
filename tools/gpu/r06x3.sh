# round 6: the new tight-loop fence (tests/test_concurrency.py) -- must fail with the formB variant library and pass with today's
export PYTHONUNBUFFERED=1
for v in formB default; do
  lib=""; [ $v != default ] && lib="E2K_LIB=$GRAFT_REPO_ROOT/tools/ab/lib/libe2k_$v.so"
  echo "== $v"; env $lib timeout 900 python -m pytest tests/test_concurrency.py -m gpu -q -p no:cacheprovider -k "back_to_back" 2>&1 | grep -E "^E  .*(Assert|bad)|passed|failed" | cut -c1-300 | head -6
done
