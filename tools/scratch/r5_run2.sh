#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_pipel.py tests/test_gpu_rounds.py tests/test_gpu_pipe.py tests/test_gpu_parity.py tests/test_gpu_big.py -x -q -m gpu > gpurun_out/r5_t3.log 2>&1 || { tail -40 gpurun_out/r5_t3.log; exit 1; }
tail -2 gpurun_out/r5_t3.log
timeout -k 10 900 python3 tools/soak.py 400 51 > gpurun_out/r05_v1_soak_400.txt 2>&1 || { tail -20 gpurun_out/r05_v1_soak_400.txt; exit 1; }
tail -1 gpurun_out/r05_v1_soak_400.txt | cut -c1-600
for W in r50_all effnet_b1 vgg16; do
  python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --layer-table gpurun_out/r05_v1_bench_${W}_layers.txt > gpurun_out/r05_v1_bench_${W}_line.json 2> gpurun_out/r05_v1_bench_${W}.err || { tail -20 gpurun_out/r05_v1_bench_${W}.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/r05_v1_bench_${W}_line.json'));print('$W',d['value'],d['ms_per_step'],d['loop_ms_per_step'],d['output_check'],d['oracle_shape_check']['mismatches'])"
done
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05_v1_bench_line.json 2> gpurun_out/r05_v1_bench.err; python3 -c "import json;d=json.load(open('gpurun_out/r05_v1_bench_line.json'));print('headline',d['value'],d['ms_per_step'])"
