#!/bin/bash
# (how to make the base library: check out the commit to compare against, `make -C quantized_neural_nets_amd/csrc`, copy
#  csrc/build/libgpfq_hip.so to csrc/stamps/libgpfq_hip_base.so -- git-ignored, travels to the GPU box -- and come back)
# same-box A/B of the pipelined / lock-step kernels: the library of HEAD (stamps/libgpfq_hip_base.so) against the working tree's
BASE=$PWD/quantized_neural_nets_amd/csrc/stamps/libgpfq_hip_base.so
SH="256,2304,26624 128,1152,26624 128,1152,93184 64,576,93184 1024,512,51200"
for i in 1 2; do
  echo "== base"; GPFQ_LIB_OVERRIDE=$BASE timeout -k 10 200 python tools/layer_bench.py $SH "GPFQ_COOP_PIPE=1" "GPFQ_COOP_PIPE=0" 2>&1 | grep "us/col"
  echo "== new";  timeout -k 10 200 python tools/layer_bench.py $SH "GPFQ_COOP_PIPE=1" "GPFQ_COOP_PIPE=0" 2>&1 | grep "us/col"
done
