#!/bin/bash
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/t41.log 2>&1; tail -3 gpurun_out/t41.log
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 --rows 3000000 --share-gpu --backend gloo --check > gpurun_out/b41.log 2>&1; tail -3 gpurun_out/b41.log | cut -c1-300
