# Weak-scaling curve on ONE node with N = 1, 2, 4, 8 GPUs (what the driver runs at round end; kept here so that the run needs
# no edits):  bash tools/gpu/scale8.sh [steps] [warmup]     -> gpurun_out/scale_N.json lines
# One process per GPU over RCCL (backend "nccl"); bench.py itself sets GPU_MAX_HW_QUEUES=8 (see its header) and reads
# RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment torch.distributed.run provides.
steps=${1:-20}; warmup=${2:-5}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus 1 --steps $steps --warmup $warmup | tee gpurun_out/scale_1.json
for n in 2 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
      bench.py --gpus $n --steps $steps --warmup $warmup --no-cpu-baseline | tee gpurun_out/scale_$n.json
done
