#!/bin/bash
# First half of a profile set (the whole of tools/profile_all.sh does not fit one 20-minute GPU call since round 5):
#   gpurun --timeout 1200 -- 'bash tools/profile_part1.sh <tag>'      plain bench lines + rocprofv3 stats + the two PMC traffic passes
# then copy gpurun_out/<tag>_pmc_traffic.json into profiles/ and run tools/profile_part2.sh <tag>.
TAG=${1:-r05_v3}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/bench_lines.sh $TAG || exit 1
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_profile_bench.log 2>&1 || { tail -20 gpurun_out/${TAG}_profile_bench.log; exit 1; }
ls gpurun_out/${TAG}_* | wc -l
