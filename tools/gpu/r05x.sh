# round 5: timeline of a replayed cfg3 step with the launch lanes on: rocprofv3 kernel trace (begin / end per dispatch) -> gaps and overlap
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05x
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/trace.log 2>&1; echo "rc=$?"
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); ls -la $f
python3 - "$f" $O/trace_small.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
keep = ['Kernel_Name', 'Start_Timestamp', 'End_Timestamp', 'Queue_Id', 'Stream_Id', 'Dispatch_Id']
keep = [k for k in keep if k in rows[0]]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-7000:]                      # the last three steps or so
with open(sys.argv[2], 'w', newline='') as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows:
        w.writerow([(r[k][:60] if k == 'Kernel_Name' else r[k]) for k in keep])
PY
grep -o '"ms_per_step": [0-9.]*' $O/trace.log
