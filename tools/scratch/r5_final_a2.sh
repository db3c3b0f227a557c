#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_workloads.py -x -q -m gpu > gpurun_out/r05_v2_gputests_workloads.log 2>&1 || { tail -40 gpurun_out/r05_v2_gputests_workloads.log; exit 1; }
tail -2 gpurun_out/r05_v2_gputests_workloads.log
bash tools/scratch/r5_final_soak.sh
