#!/bin/bash
cd $GRAFT_REPO_ROOT
export GPFQ_COOP_PIPEL=1
for P in 0 4 8 12 16 20; do
  echo "== request pause $P"
  GPFQ_PIPEL_REQUEST_PAUSE=$P DLIMIT=128 timeout -k 10 200 python3 tools/layer_bench.py "256,64,803840" "512,128,201728" "1024,256,51200" "plan=0,GPFQ_COOP_PIPEL=1,GPFQ_PIPEL_REQUEST_PAUSE=$P" 2>&1 | grep us/col
done
echo "== stamps"
for P in 0 8; do
GPFQ_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/scratch/diag/libgpfq_hip_stamps.so timeout -k 10 200 python3 tools/stamps.py GPFQ_COOP_PIPEL=1 GPFQ_PIPEL_REQUEST_PAUSE=$P 256,64,803840 512,128,201728 1024,256,51200 2>&1 | grep -v amdgpu.ids
done
