#!/bin/bash
# bench.py over every named workload (f32 output) and three of them with f16 output; one JSON line each.
# Run on the GPU box from the repo root:  tools/run_all_workloads.sh r03
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r03}
OUT=gpurun_out/${TAG}_bench_all_workloads.jsonl
mkdir -p gpurun_out
: > $OUT
for w in llama7b_q4_k_m llama7b_q2_k llama7b_q3_k llama7b_q4_k llama7b_q5_k llama7b_q6_k llama7b_q4_0 llama7b_q8_0 \
         llama7b_iq4_nl llama7b_iq4_xs tinyllama_q4_k_m q4_0_4096; do
  echo "== $w f32"
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --no-secondary 2> gpurun_out/all_$w.err | tail -1 >> $OUT
done
for w in llama7b_q4_k_m llama7b_q6_k llama7b_q2_k llama7b_q8_0 tinyllama_q4_k_m; do
  echo "== $w f16"
  timeout -k 10 300 python3 bench.py --workload $w --out f16 --no-cpu-baseline --no-secondary 2> gpurun_out/all16_$w.err | tail -1 >> $OUT
done
echo done
