#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== nccl, 2 ranks sharing GPU 0"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --rows 2000000 --share-gpu --check --cpu-sample 0 > gpurun_out/mg_nccl.log 2>&1; echo rc=$?; tail -5 gpurun_out/mg_nccl.log | cut -c1-600
echo "== gloo, 3 ranks sharing GPU 0"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 3 --steps 3 --warmup 1 --rows 1500000 --share-gpu --check --backend gloo --cpu-sample 0 > gpurun_out/mg_gloo.log 2>&1; echo rc=$?; tail -5 gpurun_out/mg_gloo.log | cut -c1-600
