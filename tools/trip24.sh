mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
SECONDS=0
python bench.py > gpurun_out/bench24_default.log 2> gpurun_out/bench24_default.err; echo "rc=$? wall=${SECONDS}s"; tail -n 1 gpurun_out/bench24_default.log
