mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 400 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --force-ddp) > gpurun_out/bench22_ddp.log 2>&1; echo "bench ddp rc=$?"; tail -n 3 gpurun_out/bench22_ddp.log | cut -c1-400
(timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 3 --no-cpu-baseline --force-ddp) > gpurun_out/bench22_trun.log 2>&1; echo "torchrun rc=$?"; tail -n 2 gpurun_out/bench22_trun.log | cut -c1-300
