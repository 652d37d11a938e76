#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=250 2>&1 | tail -2
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29515 bench.py --gpus 3 --steps 2 --warmup 1 --rows 1500000 --share-gpu --backend gloo --check --cpu-sample 0 > gpurun_out/mg_gloo4.log 2>&1; echo rc=$?; grep CHECK gpurun_out/mg_gloo4.log
