#!/bin/bash
# Group-wait sweep (Prefetch::wait) over the kernels where the packed input is a fifth of the traffic or more.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for f in Q8_0_f32 Q8_1_f32 Q8K_f32 Q4K_f16 Q6K_f16 Q5K_f16 Q8K_f16 IQ4XS_f16 IQ4NL_f16 Q4_0_f16 Q4_1_f16 Q5_0_f16 Q5_1_f16 Q8_0_f16 Q8_1_f16 ${EXTRA}; do
  timeout -k 10 300 tools/ubench/ubench 1024 7 "gw $f" > gpurun_out/gw_${f}_rep3.txt 2>&1 || exit 1
  UBENCH_REP=30 timeout -k 10 300 tools/ubench/ubench 1024 3 "gw $f" > gpurun_out/gw_${f}_rep30.txt 2>&1 || exit 1
  echo "$f done"
done
