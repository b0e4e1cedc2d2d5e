#!/bin/bash
# Per-tensor versus launch-plan file decode, kernel side from rocprofv3 (run on the GPU box from the repo root).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
TAG=${1:-r03}
for wl in tinyllama_q4_k_m llama7b_q4_k_m; do
  for mode in batch per-tensor; do
    tag=${wl}_${mode}
    echo "== $tag" 
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_fd_$tag -o fd --output-format csv -- python3 tools/measure_file_decode.py --workload $wl --mode $mode > $OUT/fd_$tag.json 2> $OUT/fd_$tag.err
    python3 profiles/summarize_file_decode.py $OUT/prof_fd_$tag $OUT/fd_$tag.json $OUT/${TAG}_file_decode_$tag.json > /dev/null
  done
  timeout -k 10 300 python3 tools/measure_file_decode.py --workload $wl --mode batch --resident > $OUT/${TAG}_resident_per_tensor_vs_plan_$wl.json 2> $OUT/fd_res_$wl.err
  tail -c 600 $OUT/${TAG}_resident_per_tensor_vs_plan_$wl.json; echo
done
