# Round 4, first GPU call: reproduce the SIGABRT of GPUTEST_r03 (test_plan_replay_with_the_default_off_switches[gpu]).
#   gpurun --timeout 1500 -- 'bash tools/gpu/r04a_abort_hunt.sh'
# 1. the driver's exact command, full stdout+stderr kept; 2. the one test in a loop under the bisection switches.
export PYTHONUNBUFFERED=1
O=gpurun_out/r04a
mkdir -p $O
T='tests/test_backbone.py::test_plan_replay_with_the_default_off_switches'
echo "== driver command" | tee $O/summary.txt
(timeout 900 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider) > $O/driver_cmd.log 2>&1
echo "driver rc=$? $(tail -1 $O/driver_cmd.log)" | tee -a $O/summary.txt
dmesg 2>/dev/null | tail -30 > $O/dmesg_after_driver.txt

loop() {   # name, count, env...
  name=$1; n=$2; shift 2
  fails=0
  for i in $(seq 1 $n); do
    (env "$@" timeout 120 python3 -m pytest "$T" -x -q -m gpu -p no:cacheprovider) > $O/${name}_$i.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "   $name run $i rc=$rc"; else rm -f $O/${name}_$i.log; fi
  done
  echo "$name: $fails / $n failed" | tee -a $O/summary.txt
}
loop default 12 E2K_DUMMY=1
if [ $fails -eq 0 ]; then
  loop default_b 25 E2K_DUMMY=1
fi
if [ $fails -gt 0 ]; then
  loop serialize 6 AMD_SERIALIZE_KERNEL=3
  loop hwq8 8 GPU_MAX_HW_QUEUES=8
  loop recast0 8 E2K_RECAST_T_ON_LANE=0
  loop lanes0 8 E2K_LANES=0
  loop lanesbwd0 6 E2K_LANES_BWD=0
  loop lanes1 6 E2K_LANES=1
  loop lanes2 6 E2K_LANES=2
fi
dmesg 2>/dev/null | tail -60 > $O/dmesg_end.txt
cat $O/summary.txt
