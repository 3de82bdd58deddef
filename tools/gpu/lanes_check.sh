mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for i in 1 2 3; do timeout 600 python -m pytest tests/test_backbone.py -m gpu -q -p no:cacheprovider -x -k "lanes or plan or persistent" 2>&1 | tail -1; done
for m in 800 960; do MTOK=$m python tools/probes/hc_concurrent.py 2>&1 | grep "mismatching" | tail -2; done
python tools/probes/victim_scan.py 2>&1 | grep -v amdgpu
E2K_LANES=3 python tools/probes/lanes_debug.py 2>&1 | grep "True, True" | tail -3
SIZE=big E2K_LANES=3 python tools/probes/lanes_debug.py 2>&1 | grep "True, True" | tail -3
