#!/bin/bash
# build the library and print the memory / wait skeleton of the pipelined reducer loop (diagnostic)
B=/root/repo/quantized_neural_nets_amd/csrc
make -C $B all 2>&1 | grep -E "problem|error|Error" | cut -c1-200
K=${1:-_ZN4gpfq19gpfq_pipe_rg1_m0_w8}
N0=$(grep -n "^$K" $B/build/gpfq_capi-hip-amdgcn-amd-amdhsa-gfx950.s | head -1 | cut -d: -f1)
sed -n "${N0},$((N0+12000))p" $B/build/gpfq_capi-hip-amdgcn-amd-amdhsa-gfx950.s | awk '{print} /s_endpgm/{exit}' > /tmp/pipe_k.s
L=$(grep -n "s_setprio" /tmp/pipe_k.s | head -1 | cut -d: -f1)
awk -v a=$L 'NR>=a && NR<=a+460' /tmp/pipe_k.s | grep -n "s_barrier\|global_load\|global_store\|s_waitcnt\|s_sleep\|ds_read\|ds_write\|s_load\|global_atomic" | head -${2:-40}
grep -A14 "Function Name: $K" $B/build/resource_usage.txt | grep -E "VGPRs:|SGPRs:|Spill|Scratch" | sed 's/remark: [^ ]* *//; s/ \[-Rpass.*//' | tr '\n' ' '; echo
