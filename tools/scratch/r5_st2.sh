#!/bin/bash
cd $GRAFT_REPO_ROOT
export GPFQ_COOP_PIPEL=1
echo "== reducer priority"
DLIMIT=128 timeout -k 10 200 python3 tools/layer_bench.py "256,64,803840" "512,128,201728" "1024,256,51200" "plan=0,GPFQ_COOP_PIPEL=1" "plan=0,GPFQ_COOP_PIPEL=1,GPFQ_PIPEL_REDUCER_PRIO0=1" 2>&1 | grep us/col
echo "== stamps per sweep wave"
for W in 100 103 104 106; do
GPFQ_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/scratch/diag/libgpfq_hip_stamps.so timeout -k 10 200 python3 tools/stamps.py GPFQ_COOP_PIPEL=1 GPFQ_COOP_PACE=$W 256,64,803840 1024,256,51200 2>&1 | grep -v "amdgpu.ids\|reducer"
done
