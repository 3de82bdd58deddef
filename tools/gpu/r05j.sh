# round 5, call 10: kernel trace of a few steady-state steps (who issues the __amd_rocclr_copyBuffer dispatches?)
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(E2K_LANES=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-launch-floor --no-optimizer-leg) > $O/kt.log 2>&1; echo "trace rc=$?"
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/copybuffer_neighbours.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
short = lambda n: n.split('(')[0][-60:]
idx = [i for i, n in enumerate(names) if 'masked_mse_fwd' in n]
print('dispatches', len(rows), 'loss kernels at', idx)
# steady state: between the last two loss kernels
a, b = idx[-2], idx[-1]
seg = names[a:b]
print('dispatches in the last step:', len(seg))
c = collections.Counter(short(n) for n in seg)
for k, v in c.most_common(12): print(v, k)
cb = [i for i in range(a, b) if 'copyBuffer' in names[i] or 'fillBuffer' in names[i]]
print('copy/fill dispatches in the last step:', len(cb))
nb = collections.Counter((short(names[i]), short(names[i - 1]), short(names[i + 1])) for i in cb)
for k, v in nb.most_common(15): print(v, k)
PY
cat $O/copybuffer_neighbours.txt | head -50
