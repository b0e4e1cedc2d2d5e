#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box from the repo root):
#   kernel-trace stats of the default bench command; FETCH_SIZE / WRITE_SIZE passes of the same launches with
#   the shipped burst-prefetch schedule and, as the control, with GGQ_NO_PREFETCH=1; kernel-trace stats and counter
#   passes of the f16-output bench; kernel-trace stats of the Llama-70B f16 resident leg and of the quantize /
#   widen / narrow / permute kernels.
# Counters are collected in passes of their own (never together with the trace).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
TAG=${1:-r03}
B="bench.py --no-cpu-baseline --streamed-steps 0"
S="--steps 2 --warmup 1 --settle-ms 0 --no-verify"
stats_csv() {   # kernel-trace dir -> profiles/<name>: this library's kernels only
python3 - "$1" "$2" <<PY
import csv, glob, sys
rows = [r for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])) if "ggq::" in r["Name"]]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_ns"])
    for r in rows:
        w.writerow([r["Name"].split("(")[0].replace("void ", ""), r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["TotalDurationNs"]])
PY
}
echo "== kernel trace, default bench"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -o kt --output-format csv -- python3 $B > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/prof_kt.err
echo "== FETCH_SIZE"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o c --output-format csv -- python3 $B $S > /dev/null 2> $OUT/prof_fetch.err
echo "== WRITE_SIZE"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write -o c --output-format csv -- python3 $B $S > /dev/null 2> $OUT/prof_write.err
python3 profiles/summarize.py $TAG $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write llama7b_q4_k_m:f32 > /dev/null
echo "== control: no burst prefetch"
export GGQ_NO_PREFETCH=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt_np -o kt --output-format csv -- python3 $B > $OUT/${TAG}_bench_no_prefetch.json 2> $OUT/prof_kt_np.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch_np -o c --output-format csv -- python3 $B $S > /dev/null 2> $OUT/prof_fetch_np.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write_np -o c --output-format csv -- python3 $B $S > /dev/null 2> $OUT/prof_write_np.err
unset GGQ_NO_PREFETCH
python3 profiles/summarize.py ${TAG}_no_prefetch $OUT/prof_kt_np $OUT/prof_fetch_np $OUT/prof_write_np llama7b_q4_k_m:f32:no_prefetch > /dev/null
echo "== f16 output: kernel trace and counters"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt_f16 -o kt --output-format csv -- python3 $B --out f16 > $OUT/${TAG}_bench_f16_under_rocprof.json 2> $OUT/prof_kt_f16.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch_f16 -o c --output-format csv -- python3 $B $S --out f16 > /dev/null 2> $OUT/prof_fetch_f16.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write_f16 -o c --output-format csv -- python3 $B $S --out f16 > /dev/null 2> $OUT/prof_write_f16.err
python3 profiles/summarize.py ${TAG}_f16 $OUT/prof_kt_f16 $OUT/prof_fetch_f16 $OUT/prof_write_f16 llama7b_q4_k_m:f16 > /dev/null
echo "== f16 output through the tile table only (scattered layout): kernel trace and counters"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt_f16s -o kt --output-format csv -- python3 $B --out f16 --layout scattered > $OUT/${TAG}_bench_f16_scattered_under_rocprof.json 2> $OUT/prof_kt_f16s.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch_f16s -o c --output-format csv -- python3 $B $S --out f16 --layout scattered > /dev/null 2> $OUT/prof_fetch_f16s.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write_f16s -o c --output-format csv -- python3 $B $S --out f16 --layout scattered > /dev/null 2> $OUT/prof_write_f16s.err
python3 profiles/summarize.py ${TAG}_f16_scattered $OUT/prof_kt_f16s $OUT/prof_fetch_f16s $OUT/prof_write_f16s llama7b_q4_k_m:f16:scattered > /dev/null
echo "== kernel trace, Llama-70B set, f16 output, resident"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt_70b -o kt --output-format csv -- python3 $B --workload llama70b_q4_k_m_iq4nl --out f16 --steps 10 > $OUT/${TAG}_bench_llama70b_f16_under_rocprof.json 2> $OUT/prof_kt_70b.err
stats_csv $OUT/prof_kt_70b profiles/${TAG}_kernel_stats_llama70b_f16.csv
echo "== kernel trace, quantize / widen / narrow / permute"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt_misc -o kt --output-format csv -- python3 tools/run_misc_kernels.py > $OUT/misc_kernels.json 2> $OUT/prof_kt_misc.err
python3 profiles/summarize_misc.py $OUT/prof_kt_misc $OUT/misc_kernels.json profiles/${TAG}_kernel_stats_misc.json > /dev/null
cp $OUT/${TAG}_bench_under_rocprof.json $OUT/${TAG}_bench_no_prefetch.json $OUT/${TAG}_bench_f16_under_rocprof.json $OUT/${TAG}_bench_f16_scattered_under_rocprof.json $OUT/${TAG}_bench_llama70b_f16_under_rocprof.json profiles/
mkdir -p $OUT/profiles_out && cp profiles/traffic.json profiles/${TAG}_*.csv profiles/${TAG}_*.json $OUT/profiles_out/
echo done
