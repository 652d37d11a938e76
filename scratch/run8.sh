#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 3 --warmup 1 --rows 3000000 --share-gpu --backend gloo --cpu-sample 0 > gpurun_out/mg_gloo2.log 2>&1; echo rc=$?; tail -1 gpurun_out/mg_gloo2.log | cut -c1-900
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 2 --steps 2 --warmup 1 --rows 3000000 --share-gpu --backend gloo --check --cpu-sample 0 > gpurun_out/mg_gloo3.log 2>&1; echo rc=$?; grep CHECK gpurun_out/mg_gloo3.log
timeout -k 10 200 python bench.py --steps 10 --warmup 2 > gpurun_out/b3.log 2>&1; tail -1 gpurun_out/b3.log
python __graft_entry__.py smoke
