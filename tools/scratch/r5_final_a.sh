#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r05_v2_gputests.log 2>&1 || { tail -40 gpurun_out/r05_v2_gputests.log; exit 1; }
tail -2 gpurun_out/r05_v2_gputests.log
