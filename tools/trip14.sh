mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p14 -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py ${1:-hc}) > $GRAFT_REPO_ROOT/gpurun_out/p14.log 2>&1; echo "rc=$?"
find /tmp/p14 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/p14_kernel_stats.csv \;
head -n 12 $GRAFT_REPO_ROOT/gpurun_out/p14_kernel_stats.csv | cut -c1-180
