#!/bin/bash
# rocprofv3 kernel statistics of one bench workload:  gpurun -- 'bash tools/profile_workload.sh effnet_b1 r02_v6'
W=${1:-effnet_b1}; TAG=${2:-r02_v6}
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_w
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_w -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-output-check > gpurun_out/${TAG}_bench_${W}_line_under_rocprof.json 2> gpurun_out/prof_w.err
cp $(find gpurun_out/prof_w -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_${W}_kernel_stats.csv
find gpurun_out/prof_w -name "*kernel_trace.csv" -delete
head -14 gpurun_out/${TAG}_bench_${W}_kernel_stats.csv | cut -c1-150
