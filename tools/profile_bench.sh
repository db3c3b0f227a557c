#!/bin/bash
# The profile set behind DESIGN.md section 7, on the GPU box:  gpurun -- 'bash tools/profile_bench.sh r02_v3'
#   1. rocprofv3 --kernel-trace --stats of bench.py (per-kernel average durations; must agree with the event timings)
#   2. two SEPARATE PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) -> tools/pmc_traffic.py (stamped with the
#      digest of the kernel sources; bench.py quotes roofline.traffic only from a summary of the sources it runs)
# Copy gpurun_out/<tag>_* into profiles/ afterwards.
TAG=${1:-r02_v3}
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-output-check > gpurun_out/${TAG}_bench_line_under_rocprof.json 2> gpurun_out/prof_stats.err
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-output-check > /dev/null 2> gpurun_out/prof_fetch.err
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-output-check > /dev/null 2> gpurun_out/prof_write.err
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -name "*.csv" | head -20
F=$(find gpurun_out/prof_fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/prof_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W > gpurun_out/${TAG}_pmc_traffic.json
cp $(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
find gpurun_out/prof_stats -name "*kernel_trace.csv" -delete
head -c 1500 gpurun_out/${TAG}_pmc_traffic.json
