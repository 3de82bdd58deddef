mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o r1 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline) > gpurun_out/rocprof_bench.log 2>&1; echo "rocprof rc=$?"
ls -R gpurun_out/prof_bench | head -30
(timeout 600 python bench.py --steps 5 --warmup 2) > gpurun_out/bench_cfg3_full.log 2>&1; echo "bench rc=$?"
tail -n 2 gpurun_out/bench_cfg3_full.log
