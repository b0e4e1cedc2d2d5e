#!/bin/bash
# quant_one<EncQ4_0, InF32> in eight placements: HIP events, kernel trace and FETCH / WRITE counters (separate
# passes).  Run on the GPU box from the repo root; the summary goes to profiles/.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/quant_placement
rm -rf $OUT
mkdir -p $OUT
timeout -k 10 200 python3 tools/probe_quant_placement.py > $OUT/plain.json 2> $OUT/plain.err
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python3 tools/probe_quant_placement.py > $OUT/kt.json 2> $OUT/kt.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o c --output-format csv -- python3 tools/probe_quant_placement.py > /dev/null 2> $OUT/fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o c --output-format csv -- python3 tools/probe_quant_placement.py > /dev/null 2> $OUT/write.err
python3 profiles/summarize_quant_placement.py $OUT/plain.json $OUT/kt $OUT/fetch $OUT/write $OUT/r03_quant_q4_0_placements.json > /dev/null
cp $OUT/r03_quant_q4_0_placements.json profiles/
echo done
