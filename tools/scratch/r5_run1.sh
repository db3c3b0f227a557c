#!/bin/bash
# round 5, first GPU call: the new / changed tests, the r50_all baseline of this box, its counter passes, the new driver benches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest "tests/test_gpu_workloads.py::test_two_rank_bench_on_one_card_over_gloo_runs_the_whole_multi_rank_script" -x -q -m gpu > gpurun_out/r5_t1.log 2>&1 || { tail -40 gpurun_out/r5_t1.log; exit 1; }
tail -3 gpurun_out/r5_t1.log
python3 bench.py --workload r50_all --steps 3 --warmup 1 --no-cpu-baseline --layer-table gpurun_out/r05_v0_bench_r50_all_layers.txt > gpurun_out/r05_v0_bench_r50_all_line.json 2> gpurun_out/r05_v0_bench_r50_all.err || { tail -20 gpurun_out/r05_v0_bench_r50_all.err; exit 1; }
echo "r50_all done"
bash tools/profile_counters.sh r05_v0 "--workload r50_all --distinct-shapes --max-cols 96" _r50_all > gpurun_out/r05_v0_profile_counters_r50_all.log 2>&1 || { tail -20 gpurun_out/r05_v0_profile_counters_r50_all.log; exit 1; }
echo "counters done"
python3 bench.py --driver effnet_b1 > gpurun_out/r05_v0_driver_effnet_b1.json 2> gpurun_out/r05_v0_driver_effnet_b1.err || { tail -30 gpurun_out/r05_v0_driver_effnet_b1.err; exit 1; }
echo "effnet driver done"
python3 bench.py --driver vgg16 > gpurun_out/r05_v0_driver_vgg16.json 2> gpurun_out/r05_v0_driver_vgg16.err || { tail -30 gpurun_out/r05_v0_driver_vgg16.err; exit 1; }
echo "vgg16 driver done"
python3 bench.py --driver r50 > gpurun_out/r05_v0_driver_r50.json 2> gpurun_out/r05_v0_driver_r50.err || { tail -30 gpurun_out/r05_v0_driver_r50.err; exit 1; }
echo "r50 driver done"
