#!/bin/bash
# Small tensors (the one-tensor 4096 x 4096 Q4_0 configuration is 16 Mi elements, a 15 us launch): workgroup shapes
# and schedules of the single-tensor decode at 16 Mi and 64 Mi elements, warm (the same input again and again: it
# stays in the 256 MiB memory-side cache) and cold (a 2 GiB read with plain loads before every sample).  Run on the GPU box.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/ubench_small.txt
: > $OUT
for mi in ${SIZES:-16 64 128 256}; do
  for cold in 0 1; do
    for sel in "sdeq" "sw Q4_0_f32" "sw Q4K_f32" "sw Q6K_f32"; do
      echo "== $mi Mi elements, cold=$cold, $sel" >> $OUT
      UBENCH_COLD=$cold UBENCH_REP=1 timeout -k 10 200 tools/ubench/ubench $mi 31 "$sel" >> $OUT 2>&1 || exit 1
    done
  done
done
echo done
