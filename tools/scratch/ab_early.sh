#!/bin/bash
# experiment: the gather requested in the phase its group is PUBLISHED in (a three-phase chain inside the four-phase cycle)
EARLY=$PWD/quantized_neural_nets_amd/csrc/stamps/libgpfq_hip_early.so
BASE=$PWD/quantized_neural_nets_amd/csrc/stamps/libgpfq_hip_base.so
SH="256,2304,26624 128,1152,26624 128,1152,93184 64,576,93184 1024,512,51200"
echo "== base";  GPFQ_LIB_OVERRIDE=$BASE timeout -k 10 200 python tools/layer_bench.py $SH "GPFQ_COOP_PIPE=1" 2>&1 | grep "us/col" | cut -c1-24,60-200
echo "== early (request delay 0, 2, 4, 8, 12 x 64 clocks)"
GPFQ_LIB_OVERRIDE=$EARLY timeout -k 10 400 python tools/layer_bench.py $SH "GPFQ_COOP_PIPE=1,GPFQ_COOP_PACE=0" "GPFQ_COOP_PIPE=1,GPFQ_COOP_PACE=2" "GPFQ_COOP_PIPE=1,GPFQ_COOP_PACE=4" "GPFQ_COOP_PIPE=1,GPFQ_COOP_PACE=8" "GPFQ_COOP_PIPE=1,GPFQ_COOP_PACE=12" 2>&1 | grep "us/col" | cut -c1-24,48-64,80-200
