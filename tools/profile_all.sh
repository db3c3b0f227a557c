#!/bin/bash
# The whole profile set of a build, on the GPU box:   gpurun --timeout 1150 -- 'bash tools/profile_all.sh r03_v1'
# Writes gpurun_out/<tag>_*; copy them into profiles/ afterwards (that directory is what is committed and judged).
#   1. bench lines + per-layer tables of the headline and the secondary workloads, and the emulated per-rank steps of
#      2 / 4 / 8 GPUs (tools/bench_lines.sh: plain runs, these are the numbers)
#   2. rocprofv3 --kernel-trace --stats of the headline, two PMC passes for the HBM-side traffic (tools/profile_bench.sh)
#   3. the counter passes behind roofline_issue / the measured roofline_l2 (tools/profile_counters.sh)
#   4. rocprofv3 kernel statistics of r50_all and effnet_b1 (tools/profile_workload.sh)
# Since round 5 the whole set takes ~25 minutes, more than one 20-minute GPU call: use tools/profile_part1.sh and
# tools/profile_part2.sh (the same steps in two calls; part 2 copies the counter summaries into profiles/ on the box and re-runs the
# lines that quote them).  This script remains for a box without that limit.
TAG=${1:-r03_v1}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/bench_lines.sh $TAG || exit 1
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_profile_bench.log 2>&1 || exit 1
bash tools/profile_counters.sh $TAG > gpurun_out/${TAG}_profile_counters.log 2>&1 || exit 1
bash tools/profile_workload.sh r50_all $TAG > /dev/null 2>&1 || exit 1
bash tools/profile_workload.sh effnet_b1 $TAG > /dev/null 2>&1 || exit 1
# round 5: the counter passes of the secondary workload whose kernels the headline does not launch (every distinct layer shape of
# ResNet-50, full N / m / groups, 96 input features per group), and quantize_network() itself on the four architectures
bash tools/profile_counters.sh $TAG "--workload r50_all --distinct-shapes --max-cols 96" _r50_all > gpurun_out/${TAG}_profile_counters_r50_all.log 2>&1 || exit 1
for D in r18 r50 vgg16 effnet_b1; do
  python3 bench.py --driver $D > gpurun_out/${TAG}_driver_${D}.json 2> gpurun_out/${TAG}_driver_${D}.err || exit 1
done
# bench.py quotes a PMC summary only from profiles/: copy <tag>_pmc_*.json there and run tools/bench_lines.sh once more for
# the headline line that carries roofline.traffic, roofline_issue and the measured roofline_l2
ls -la gpurun_out/${TAG}_* | awk '{print $5, $9}'
