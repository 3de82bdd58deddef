# round 6: which tensor differs when plans + lanes disagree with the single-stream schedule
export PYTHONUNBUFFERED=1
for v in "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1"; do
for i in 1 2 3; do
  echo "== $v run $i"; env $v REPS=4 timeout 300 python tools/probes/lanes_race.py 2>&1 | tail -14 | cut -c1-700
done; done
