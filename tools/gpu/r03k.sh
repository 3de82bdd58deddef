# Round 3, GPU call 11: why did the host launch floor go from 20 to 29 ms (wall 41)?  launch_floor probe with the round's switches off one by one,
# and its per-call profile
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for v in "default:" "no_group:E2K_WGRAD_GROUP=0" "no_dual:E2K_WGRAD_DUAL=0" "no_group_no_dual:E2K_WGRAD_GROUP=0 E2K_WGRAD_DUAL=0" "no_stage:E2K_GEMM_FLAGS=64" "q4:GPU_MAX_HW_QUEUES=4"; do
  name=${v%%:*}; envs=${v#*:}
  (env $envs timeout 120 python tools/probes/launch_floor.py) > gpurun_out/lf_$name.log 2>&1
  echo "[$name] $(grep -v amdgpu.ids gpurun_out/lf_$name.log | tail -n 1 | cut -c1-330)"
done
cd /tmp && export TMPDIR=/tmp
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lf -- python $GRAFT_REPO_ROOT/tools/probes/launch_floor.py) > $GRAFT_REPO_ROOT/gpurun_out/prof_lf.log 2>&1
find /tmp/prof_lf -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_lf_kernel_stats.csv \;
head -n 12 $GRAFT_REPO_ROOT/gpurun_out/prof_lf_kernel_stats.csv | cut -c1-170
