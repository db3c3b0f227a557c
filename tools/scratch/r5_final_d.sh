#!/bin/bash
cd $GRAFT_REPO_ROOT
TAG=r05_v2
python3 bench.py --steps 20 --warmup 5 --layer-table gpurun_out/${TAG}_bench_layers.txt > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
echo "headline done"
for W in effnet_b1 vgg16 r18; do
  python3 bench.py --workload $W --steps 3 --warmup 1 --layer-table gpurun_out/${TAG}_bench_${W}_layers.txt > gpurun_out/${TAG}_bench_${W}_line.json 2> gpurun_out/${TAG}_bench_${W}.err || { tail -20 gpurun_out/${TAG}_bench_${W}.err; exit 1; }
done
python3 bench.py --workload r50_all --steps 3 --warmup 1 --layer-table gpurun_out/${TAG}_bench_r50_all_layers.txt > gpurun_out/${TAG}_bench_r50_all_line.json 2> gpurun_out/${TAG}_bench_r50_all.err || { tail -20 gpurun_out/${TAG}_bench_r50_all.err; exit 1; }
echo "workload lines done"
bash tools/profile_workload.sh r50_all $TAG > /dev/null 2>&1 || { echo "profile_workload r50_all failed"; exit 1; }
bash tools/profile_workload.sh effnet_b1 $TAG > /dev/null 2>&1 || { echo "profile_workload effnet failed"; exit 1; }
echo "kernel stats done"
for D in r18 r50 vgg16 effnet_b1; do
  python3 bench.py --driver $D > gpurun_out/${TAG}_driver_${D}.json 2> gpurun_out/${TAG}_driver_${D}.err || { tail -20 gpurun_out/${TAG}_driver_${D}.err; exit 1; }
done
echo "drivers done"
