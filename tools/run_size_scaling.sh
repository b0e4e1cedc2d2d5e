#!/bin/bash
# Size dependence of the single-run decode (see tools/probe_size_scaling.py), several schedules, one box.
cd "$(dirname "$0")/.."
S="2048,4096,8192,16384,32768"
T=${1:-12}
O=${2:-f16}
timeout -k 10 200 python3 tools/probe_size_scaling.py $O $S $T || exit 1
GGQ_HYBRID_MIN_ELEMS=0 timeout -k 10 200 python3 tools/probe_size_scaling.py $O $S $T || exit 1
for sch in "1,2048,4" "1,1280,6" "0,0,6" "1,4096,6" "768,1536,0" "768,1536,6"; do
  GGQ_PREFETCH=$sch timeout -k 10 200 python3 tools/probe_size_scaling.py $O $S $T || exit 1
done
