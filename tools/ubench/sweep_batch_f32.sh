#!/bin/bash
# period x lead sweep of the batch kernel, f32 output, every format (10 launches per sample); prints the best six per format
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for f in Q2K Q3K Q4K Q5K Q6K Q8K IQ4XS IQ4NL Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q8_1; do
  UBENCH_REP=10 timeout -k 10 400 tools/ubench/ubench 1024 7 "swb ${f}_f32" > gpurun_out/swb_${f}_f32.txt 2>&1 || exit 1
  echo "== $f"; grep "swb" gpurun_out/swb_${f}_f32.txt | sort -k6 -n | head -6
done
