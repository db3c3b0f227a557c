#!/bin/bash
# Second half of a profile set:   gpurun --timeout 1200 -- 'bash tools/profile_part2.sh <tag>'
# counter passes (headline, then every distinct ResNet-50 layer shape), their summaries copied into profiles/ ON THE BOX so that the
# lines that follow quote them, the final headline / r50_all lines, rocprofv3 kernel statistics of the two large secondary
# workloads, quantize_network() itself on the four architectures.
TAG=${1:-r05_v3}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_counters.sh $TAG > gpurun_out/${TAG}_profile_counters.log 2>&1 || { tail -20 gpurun_out/${TAG}_profile_counters.log; exit 1; }
bash tools/profile_counters.sh $TAG "--workload r50_all --distinct-shapes --max-cols 96" _r50_all > gpurun_out/${TAG}_profile_counters_r50_all.log 2>&1 || { tail -20 gpurun_out/${TAG}_profile_counters_r50_all.log; exit 1; }
cp gpurun_out/${TAG}_pmc_counters.json gpurun_out/${TAG}_pmc_counters_r50_all.json profiles/
echo "counters done"
python3 bench.py --steps 20 --warmup 5 --layer-table gpurun_out/${TAG}_bench_layers.txt > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
python3 bench.py --workload r50_all --steps 3 --warmup 1 --layer-table gpurun_out/${TAG}_bench_r50_all_layers.txt > gpurun_out/${TAG}_bench_r50_all_line.json 2> gpurun_out/${TAG}_bench_r50_all.err || { tail -20 gpurun_out/${TAG}_bench_r50_all.err; exit 1; }
echo "final lines done"
bash tools/profile_workload.sh r50_all $TAG > /dev/null 2>&1 || { echo "profile_workload r50_all failed"; exit 1; }
bash tools/profile_workload.sh effnet_b1 $TAG > /dev/null 2>&1 || { echo "profile_workload effnet_b1 failed"; exit 1; }
for D in r18 r50 vgg16 effnet_b1; do
  python3 bench.py --driver $D > gpurun_out/${TAG}_driver_${D}.json 2> gpurun_out/${TAG}_driver_${D}.err || { tail -20 gpurun_out/${TAG}_driver_${D}.err; exit 1; }
done
echo "drivers done"
