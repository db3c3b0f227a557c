#!/bin/bash
# The whole profile set of a build, on the GPU box:   gpurun --timeout 1150 -- 'bash tools/profile_all.sh r03_v1'
# Writes gpurun_out/<tag>_*; copy them into profiles/ afterwards (that directory is what is committed and judged).
#   1. bench lines + per-layer tables of the headline and the secondary workloads (plain runs: these are the numbers)
#   2. rocprofv3 --kernel-trace --stats of the headline, two PMC passes for the HBM-side traffic (tools/profile_bench.sh)
#   3. the counter passes behind roofline_issue / the measured roofline_l2 (tools/profile_counters.sh)
#   4. rocprofv3 kernel statistics of r50_all and effnet_b1 (tools/profile_workload.sh)
#   5. the emulated per-rank steps of 2 / 4 / 8 GPUs (bench.py --emulate-world)
TAG=${1:-r03_v1}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --steps 20 --warmup 5 --layer-table gpurun_out/${TAG}_bench_layers.txt > gpurun_out/${TAG}_bench_line_pre.json 2> gpurun_out/${TAG}_bench.err || exit 1
for W in r50_all effnet_b1 vgg16 r18; do
  python3 bench.py --workload $W --steps 3 --warmup 1 --layer-table gpurun_out/${TAG}_bench_${W}_layers.txt > gpurun_out/${TAG}_bench_${W}_line.json 2> gpurun_out/${TAG}_bench_${W}.err || exit 1
done
python3 bench.py --capture --steps 5 --warmup 2 --no-cpu-baseline --layer-table gpurun_out/${TAG}_bench_capture_layers.txt > gpurun_out/${TAG}_bench_capture_line.json 2> gpurun_out/${TAG}_bench_capture.err || exit 1
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_profile_bench.log 2>&1 || exit 1
bash tools/profile_counters.sh $TAG > gpurun_out/${TAG}_profile_counters.log 2>&1 || exit 1
bash tools/profile_workload.sh r50_all $TAG > /dev/null 2>&1 || exit 1
bash tools/profile_workload.sh effnet_b1 $TAG > /dev/null 2>&1 || exit 1
for N in 2 4 8; do
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --emulate-world $N > gpurun_out/${TAG}_emulated_world${N}_line.json 2> /dev/null || exit 1
  python3 bench.py --workload r50_all --steps 2 --warmup 1 --no-cpu-baseline --emulate-world $N > gpurun_out/${TAG}_emulated_world${N}_r50_all_line.json 2> /dev/null || exit 1
done
# the headline line once more, now that the counter summaries of THIS build exist next to it (bench.py quotes a summary only
# from profiles/: the line committed as <tag>_bench_line.json is produced after the summaries have been copied there)
ls -la gpurun_out/${TAG}_* | awk '{print $5, $9}'
