#!/bin/bash
# N = 2 and 4 rehearsal of bench.py on the one GPU of the box (gloo for the collectives, every rank on cuda:0):
# the sharded table must equal the single-GPU table (--check)
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) bench.py --gpus $n --backend gloo --share-gpu --check --rows 1500000 --steps 3 --warmup 1 2>&1 | grep -v "^W\|Warning\|warn" | tail -2 | cut -c1-260 || exit 1
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 2 --backend gloo --share-gpu --check --workload cfg3 --rows 6000000 --steps 3 --warmup 1 2>&1 | grep -v "^W\|Warning\|warn" | tail -2 | cut -c1-260
