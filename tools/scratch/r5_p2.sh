#!/bin/bash
cd $GRAFT_REPO_ROOT
GPFQ_PIPEL_GROUPS=2 timeout -k 10 300 python3 -m pytest tests/test_gpu_pipel.py -x -q -m gpu > gpurun_out/r5_t5.log 2>&1 || { tail -30 gpurun_out/r5_t5.log; exit 1; }
tail -2 gpurun_out/r5_t5.log
for P in 4 8 12; do
DLIMIT=128 timeout -k 10 300 python3 tools/layer_bench.py "256,64,803840" "1024,512,201728" "1024,256,51200" "2048,512,13312" "64,147,263168" "plan=0,GPFQ_COOP_PIPEL=1" "plan=0,GPFQ_COOP_PIPEL=1,GPFQ_PIPEL_GROUPS=2,GPFQ_PIPEL_REQUEST_PAUSE=$P" 2>&1 | grep us/col | awk '{print $1,$3,$4,$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}'
done
