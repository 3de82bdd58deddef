mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d /tmp/pmc_tr -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graphs) > $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic.log 2>&1; echo "pmc rc=$?"
find /tmp/pmc_tr -name "*counter_collection.csv" -exec cp {} /tmp/cc.csv \;
ls -la /tmp/cc.csv
python - <<'PY'
import csv, collections, json, os, re
rows = csv.DictReader(open('/tmp/cc.csv'))
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in rows:
    m = re.search(r'(gemm_nt_glds_kernel|gemm_nt_fixup_kernel|gemm_tn_glds_kernel|tn_reduce_kernel|attn_fwd_kernel|attn_bwd_dq_kernel|attn_bwd_dkv_kernel|hc_fwd_kernel|hc_bwd_kernel)', r['Kernel_Name'])
    if not m: continue
    a = agg[m.group(1)][r['Counter_Name']]
    a[0] += float(r['Counter_Value']); a[1] += 1
out = {k: {c: {'sum': v[0], 'dispatches': v[1], 'avg': v[0] / max(v[1], 1)} for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_traffic.json', 'w'), indent=1)
for k, d in out.items(): print(k, {c: (round(v['avg'], 1), v['dispatches']) for c, v in d.items()})
PY
