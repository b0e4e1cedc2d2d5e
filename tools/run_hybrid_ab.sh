#!/bin/bash
# A/B of the hybrid launch plans on the benchmark legs, one box, alternating settings:
#   scattered  --layout scattered               every packed tensor its own allocation: every tensor is a run
#   table      GGQ_HYBRID_MIN_ELEMS=0           typed layout, but every run through the tile table (deq_batch)
#   typed      default                          typed layout: one run per type; runs with a group-start schedule
#                                               get one launch of the single-tensor kernel (deq_one)
#   any        GGQ_HYBRID_ANY=1                 typed layout, every run above the limit gets its own launch
# Run on the GPU box from the repo root.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/hybrid_ab
rm -rf $OUT
mkdir -p $OUT
B="bench.py --no-cpu-baseline --streamed-steps 0"
run() {  # name, workload, out, steps, layout, env...
  local name=$1 wl=$2 out=$3 steps=$4 layout=$5
  shift 5
  echo "== $name"
  env "$@" timeout -k 10 400 python3 $B --workload $wl --out $out --steps $steps --layout $layout > $OUT/$name.json 2> $OUT/err.txt
}
for rep in 1 2; do
  run 7b_f16_scattered_$rep llama7b_q4_k_m f16 20 scattered GGQ_X=0
  run 7b_f16_table_$rep llama7b_q4_k_m f16 20 typed GGQ_HYBRID_MIN_ELEMS=0
  run 7b_f16_typed_$rep llama7b_q4_k_m f16 20 typed GGQ_X=0
  run 7b_f32_scattered_$rep llama7b_q4_k_m f32 20 scattered GGQ_X=0
  run 7b_f32_typed_$rep llama7b_q4_k_m f32 20 typed GGQ_X=0
  run 7b_f32_any_$rep llama7b_q4_k_m f32 20 typed GGQ_HYBRID_ANY=1
  run tiny_f16_table_$rep tinyllama_q4_k_m f16 40 typed GGQ_HYBRID_MIN_ELEMS=0
  run tiny_f16_typed_$rep tinyllama_q4_k_m f16 40 typed GGQ_X=0
  run 7b_q8_0_f32_table_$rep llama7b_q8_0 f32 20 typed GGQ_HYBRID_MIN_ELEMS=0
  run 7b_q8_0_f32_typed_$rep llama7b_q8_0 f32 20 typed GGQ_X=0
done
for rep in 1 2; do
  run 70b_f16_scattered_$rep llama70b_q4_k_m_iq4nl f16 10 scattered GGQ_X=0
  run 70b_f16_table_$rep llama70b_q4_k_m_iq4nl f16 10 typed GGQ_HYBRID_MIN_ELEMS=0
  run 70b_f16_typed_$rep llama70b_q4_k_m_iq4nl f16 10 typed GGQ_X=0
done
python3 - <<PY
import glob, json
for f in sorted(glob.glob("$OUT/*.json")):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "ms/step", r["ms_per_step"], "GB/s", r["value"], "alg GB/s", r["algorithmic_GBps"], "frac", r["roofline"]["frac"],
          [(k["type"], k["launches"], k["avg_ms"]) for k in r["kernels"]], "verified", r["verified"]["ok"])
PY
