# Round 3, GPU call 4: staged (whole-line) NT epilogue and fragment-order TN partials -- hardware tests, per-shape A/B, bench A/B;
# kernel-trace summaries of the plain and the forced data-parallel step (what the 13 ms of --force-ddp are)
tag=${1:-r03d}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 400 python -m pytest tests/test_kernels_gemm.py tests/test_backbone.py -m gpu -q -p no:cacheprovider -x) > gpurun_out/pytest_${tag}_gemm.log 2>&1; echo "pytest gemm+backbone rc=$?"; tail -n 4 gpurun_out/pytest_${tag}_gemm.log
(timeout 300 python tools/probes/gemm_epilogue_ab.py) > gpurun_out/gemm_epilogue_ab_$tag.log 2>&1; echo "epilogue probe rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_epilogue_ab_$tag.log | cut -c1-330 | tail -n 26
run() { t=$1; shift; (timeout 400 "$@") > gpurun_out/bench_${tag}_$t.log 2>&1; echo "[$t] rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${tag}_$t.log) $(grep -o '"frac": [0-9.]*' gpurun_out/bench_${tag}_$t.log | head -1) $(grep -o '"gemm_tn_bf16": {[^}]*}' gpurun_out/bench_${tag}_$t.log | head -1)"; tail -n 3 gpurun_out/bench_${tag}_$t.log | grep -v '^{' | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -n 2; }
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-launch-floor"
run plan $B
run no_stage env E2K_GEMM_FLAGS=64 $B
run plan_again $B
cd /tmp && export TMPDIR=/tmp
for v in plain ddp; do
  extra=""; [ $v = ddp ] && extra="--force-ddp"
  (timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-floor $extra) > $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_$v.log 2>&1; echo "prof $v rc=$?"
  find /tmp/prof_$v -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_${v}_kernel_stats.csv \;
  grep -o '"ms_per_step": [0-9.]*' $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_$v.log | head -1
done
python - <<'P'
import csv, os
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/'
def load(v):
    return {r['Name']: (int(r['Calls']), int(r['TotalDurationNs'])) for r in csv.DictReader(open(root + f'prof_r03d_{v}_kernel_stats.csv'))}
a, b = load('plain'), load('ddp')
rows = sorted(((b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1], k) for k in set(a) | set(b)), reverse=True)
print('largest kernel-time differences ddp - plain (ms over 4 steps):')
for d, k in rows[:8]:
    print(f'  {d / 1e6:8.2f}  calls {a.get(k, (0, 0))[0]} -> {b.get(k, (0, 0))[0]}  {k[:110]}')
print('  total', sum(v[1] for v in a.values()) / 1e6, '->', sum(v[1] for v in b.values()) / 1e6)
P
