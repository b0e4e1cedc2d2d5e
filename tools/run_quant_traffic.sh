#!/bin/bash
# HBM traffic of the quantizer kernels from the PMC counters (run on the GPU box from the repo root): kernel trace and
# separate FETCH_SIZE / WRITE_SIZE passes of tools/run_quantize_once.py (256 Mi elements per launch), merged into
# profiles/traffic.json under "quantize_f32_256Mi:<encoder>" / "quantize_f16_256Mi:<encoder>".
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
TAG=${1:-r03}
for in_ty in f32 f16; do
  flag=""; [ $in_ty = f16 ] && flag="--f16"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt_q$in_ty -o kt --output-format csv -- python3 tools/run_quantize_once.py $flag > /dev/null 2> $OUT/prof_kt_q$in_ty.err
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch_q$in_ty -o c --output-format csv -- python3 tools/run_quantize_once.py $flag > /dev/null 2> $OUT/prof_fetch_q$in_ty.err
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write_q$in_ty -o c --output-format csv -- python3 tools/run_quantize_once.py $flag > /dev/null 2> $OUT/prof_write_q$in_ty.err
  python3 profiles/summarize.py ${TAG}_quantize_$in_ty $OUT/prof_kt_q$in_ty $OUT/prof_fetch_q$in_ty $OUT/prof_write_q$in_ty quantize_${in_ty}_256Mi > /dev/null
done
mkdir -p $OUT/profiles_out && cp profiles/traffic.json profiles/${TAG}_quantize_*kernel_stats.csv $OUT/profiles_out/
python3 - <<PY
import json
d = json.load(open("profiles/traffic.json"))
for k, v in d.items():
    if k.startswith("quantize_") and v.get("round", "").startswith("$TAG"):
        print(k, {x: v[x] for x in v if "bytes" in x})
PY
