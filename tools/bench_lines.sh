#!/bin/bash
# The plain bench lines of a build (no profiler attached), on the GPU box:   gpurun -- 'bash tools/bench_lines.sh r03_v5'
# headline (+ per-layer table), the secondary workloads, the capture path, the emulated per-rank steps of 2 / 4 / 8 GPUs.
# bench.py quotes PMC summaries only from profiles/ and only with the digest of the sources it runs: with
# profiles/<tag>_pmc_{traffic,counters}.json of THIS build in place the headline line carries roofline.traffic,
# roofline_issue and the measured roofline_l2.
TAG=${1:-r03_v1}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --steps 20 --warmup 5 --layer-table gpurun_out/${TAG}_bench_layers.txt > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err || exit 1
for W in r50_all effnet_b1 vgg16 r18; do
  python3 bench.py --workload $W --steps 3 --warmup 1 --layer-table gpurun_out/${TAG}_bench_${W}_layers.txt > gpurun_out/${TAG}_bench_${W}_line.json 2> gpurun_out/${TAG}_bench_${W}.err || exit 1
done
python3 bench.py --capture --steps 5 --warmup 2 --no-cpu-baseline --layer-table gpurun_out/${TAG}_bench_capture_layers.txt > gpurun_out/${TAG}_bench_capture_line.json 2> gpurun_out/${TAG}_bench_capture.err || exit 1
for N in 2 4 8; do
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --emulate-world $N > gpurun_out/${TAG}_emulated_world${N}_line.json 2> /dev/null || exit 1
  python3 bench.py --workload r50_all --steps 2 --warmup 1 --no-cpu-baseline --emulate-world $N > gpurun_out/${TAG}_emulated_world${N}_r50_all_line.json 2> /dev/null || exit 1
done
