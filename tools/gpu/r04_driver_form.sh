# the driver's round-end sequence on a fresh box: build check is done on CPU; here smoke(), the GPU tests, the default bench
#   gpurun --timeout 1500 -- 'bash tools/gpu/r04_driver_form.sh'
export PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_driver_form
mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log | cut -c1-200)"
(timeout 1000 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_gpu.log)"
(timeout 300 python bench.py) > $O/bench_default.log 2>&1; echo "bench rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_default.log | head -1)"
