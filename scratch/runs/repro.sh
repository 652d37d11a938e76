#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for e in "A=1"; do
  echo "== $e"; env $e timeout -k 10 120 python scratch/repro_211040.py 2>&1 | grep -v amdgpu.ids | tail -14
done
