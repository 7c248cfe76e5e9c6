#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== bench 1 rank via torchrun (nccl, world 1)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 4 --warmup 1 --cpu-seconds 0 --nodes 200000 --paths 20 2>&1 | grep -v amdgpu.ids | tail -3
echo "== bench 2 ranks sharing the GPU (gloo)"
PGSGD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 4 --warmup 1 --nodes 200000 --paths 20 --stress 2>&1 | grep -v amdgpu.ids | tail -4
echo "== bench 4 ranks sharing the GPU (gloo)"
PGSGD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 4 --steps 27 --warmup 3 --nodes 200000 --paths 20 --stress 2>&1 | grep -v amdgpu.ids | tail -4
echo "== single rank same size for comparison"
timeout 600 python bench.py --steps 27 --warmup 3 --nodes 200000 --paths 20 --stress --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -2
