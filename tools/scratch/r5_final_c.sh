#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r05_v2 > gpurun_out/r05_v2_profile_bench.log 2>&1 || { tail -20 gpurun_out/r05_v2_profile_bench.log; exit 1; }
echo "profile_bench done"
bash tools/profile_counters.sh r05_v2 > gpurun_out/r05_v2_profile_counters.log 2>&1 || { tail -20 gpurun_out/r05_v2_profile_counters.log; exit 1; }
echo "counters done"
bash tools/profile_counters.sh r05_v2 "--workload r50_all --distinct-shapes --max-cols 96" _r50_all > gpurun_out/r05_v2_profile_counters_r50_all.log 2>&1 || { tail -20 gpurun_out/r05_v2_profile_counters_r50_all.log; exit 1; }
echo "r50_all counters done"
