#!/bin/bash
# (how to make the base library: check out the commit to compare against, `make -C quantized_neural_nets_amd/csrc`, copy
#  csrc/build/libgpfq_hip.so to csrc/stamps/libgpfq_hip_base.so -- git-ignored, travels to the GPU box -- and come back)
# same-box A/B inside the headline bench: lock-step XCD-local publishing off / on / off / on, per-layer loop times of layer2.{1,2,3}.conv2
for i in 1 2; do
  for L in 0 1; do
    GPFQ_COOP_LOCAL=$L python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-output-check --layer-table gpurun_out/ab_layers_$L.txt > gpurun_out/ab_line_$L.json 2>/dev/null
    echo "LOCAL=$L $(python3 -c "import json;d=json.load(open('gpurun_out/ab_line_$L.json'));print(d['value'], d['loop_ms_per_step'])") $(grep 'layer2.[123]' gpurun_out/ab_layers_$L.txt | sed 's/.*loop *\([0-9.]* ms\).*/\1/' | tr '\n' ' ')"
  done
done
BASE=$PWD/quantized_neural_nets_amd/csrc/stamps/libgpfq_hip_base.so
GPFQ_LIB_OVERRIDE=$BASE python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-output-check --layer-table gpurun_out/ab_layers_b.txt > gpurun_out/ab_line_b.json 2>/dev/null
echo "BASE $(python3 -c "import json;d=json.load(open('gpurun_out/ab_line_b.json'));print(d['value'], d['loop_ms_per_step'])") $(grep 'layer2.[123]' gpurun_out/ab_layers_b.txt | sed 's/.*loop *\([0-9.]* ms\).*/\1/' | tr '\n' ' ')"
