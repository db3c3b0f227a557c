#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests/test_gpu_pipel.py -x -q -m gpu -k "two_group" > gpurun_out/r5_t6.log 2>&1 || { tail -30 gpurun_out/r5_t6.log; exit 1; }
tail -2 gpurun_out/r5_t6.log
for P in 4 8 12 16; do
DLIMIT=128 timeout -k 10 300 python3 tools/layer_bench.py "1024,256,51200" "2048,1024,51200" "256,1024,51200" "512,128,201728" "128,512,201728" "2048,512,13312" "512,2048,13312" "256,2304,26624" "128,1152,93184" "plan=0,GPFQ_COOP_PIPE2=0,GPFQ_COOP_PIPEL=0" "plan=0,GPFQ_COOP_PIPE2=1,GPFQ_COOP_PIPEL=0,GPFQ_PIPE2_REQUEST_PAUSE=$P" 2>&1 | grep us/col | awk '{print $1,$3,$4,$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}'
done
