# round 4, call c: new defaults on hardware + A/B of this round's schedule changes + CU-mask lanes
#   gpurun --timeout 1500 -- 'bash tools/gpu/r04c.sh'
export PYTHONUNBUFFERED=1
O=gpurun_out/r04c
mkdir -p $O
(timeout 600 python -m pytest tests/test_backbone.py tests/test_kernels_hc.py tests/test_kernels_attn.py tests/test_abi.py -x -q -m gpu -p no:cacheprovider -k "not golden and not widths") > $O/pytest_subset.log 2>&1
echo "pytest subset rc=$? $(tail -1 $O/pytest_subset.log)"
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  (env "${envs[@]}" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer-leg --no-launch-floor "$@") > $O/bench_$name.log 2>&1
  echo "$name rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"launches_per_step": [0-9]*' $O/bench_$name.log | head -1)"
  grep '^{' $O/bench_$name.log | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({'name':'$name','ms_per_step':j['ms_per_step'],'cu_masks':j.get('cu_masks'),'lane_ms':j.get('lane_ms_per_step')}))" >> $O/ab.jsonl 2>/dev/null
}
run base E2K_X=1 --
run base2 E2K_X=1 --
run nozero E2K_ZERO_GRADS_ON_LANE=0 --
run nobatch E2K_BATCH_REDUCES=0 --
run text64 E2K_LANE_CUS=0:64, --
run text128 E2K_LANE_CUS=0:128, --
run text64_wgrad64 E2K_LANE_CUS=0:64,64:64 --
run main192_text64 E2K_LANE_CUS=0:64, -- --main-cus 64:192
run base3 E2K_X=1 --
cat $O/ab.jsonl
