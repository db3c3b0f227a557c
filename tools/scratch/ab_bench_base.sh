#!/bin/bash
# (how to make the base library: check out the commit to compare against, `make -C quantized_neural_nets_amd/csrc`, copy
#  csrc/build/libgpfq_hip.so to csrc/stamps/libgpfq_hip_base.so -- git-ignored, travels to the GPU box -- and come back)
# same-box A/B inside bench.py: the library of the last profile set (stamps/libgpfq_hip_base.so) against the working tree's
BASE=$PWD/quantized_neural_nets_amd/csrc/stamps/libgpfq_hip_base.so
run() { # label, workload args...
  L=$1; shift
  python3 bench.py "$@" --no-cpu-baseline --no-output-check --layer-table gpurun_out/ab_layers_$L.txt > gpurun_out/ab_line_$L.json 2>/dev/null
  echo "$L $(python3 -c "import json;d=json.load(open('gpurun_out/ab_line_$L.json'));print(d['value'], d['ms_per_step'], d['loop_ms_per_step'])")"
}
for i in 1 2; do
  GPFQ_LIB_OVERRIDE=$BASE run base$i --steps 10 --warmup 3
  run new$i --steps 10 --warmup 3
done
GPFQ_LIB_OVERRIDE=$BASE run base_r50 --workload r50_all --steps 3 --warmup 1
run new_r50 --workload r50_all --steps 3 --warmup 1
GPFQ_LIB_OVERRIDE=$BASE run base_eff --workload effnet_b1 --steps 3 --warmup 1
run new_eff --workload effnet_b1 --steps 3 --warmup 1
paste <(sed 's/.*loop *\([0-9.]*\) ms.*/\1/' gpurun_out/ab_layers_base2.txt) <(sed 's/.*loop *\([0-9.]*\) ms.*/\1/' gpurun_out/ab_layers_new2.txt) <(cut -c1-60 gpurun_out/ab_layers_base2.txt)
