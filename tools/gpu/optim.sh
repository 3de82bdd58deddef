# optimizer-side kernels: gpu tests + cfg3-scale timing.   gpurun --timeout 500 -- 'bash tools/gpu/optim.sh'
export PYTHONUNBUFFERED=1
(timeout 200 python -m pytest tests/test_optim.py -m gpu -q -p no:cacheprovider -x) 2>&1 | tail -2
(timeout 200 python tools/bench_optim.py) 2>&1 | tail -3
