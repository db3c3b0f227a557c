#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 tools/soak.py 1500 2025 > gpurun_out/r05_v2_soak_1500.txt 2>&1 || { tail -20 gpurun_out/r05_v2_soak_1500.txt; exit 1; }
tail -1 gpurun_out/r05_v2_soak_1500.txt | cut -c1-400
timeout -k 10 300 python3 tools/soak_capture.py 200 > gpurun_out/r05_v2_soak_capture_200.txt 2>&1 || { tail -20 gpurun_out/r05_v2_soak_capture_200.txt; exit 1; }
tail -1 gpurun_out/r05_v2_soak_capture_200.txt | cut -c1-300
