#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box from the repo root):
#   kernel-trace stats of the default bench command; FETCH_SIZE / WRITE_SIZE passes of the same launches with
#   the shipped burst-prefetch schedule and, as the control, with GGQ_NO_PREFETCH=1; kernel-trace stats of the
#   f16-output bench and of the quantize / widen / narrow / permute kernels.
# Counters are collected in passes of their own (never together with the trace).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
TAG=${1:-r02}
B="bench.py --no-cpu-baseline --streamed-steps 0"
echo "== kernel trace, default bench"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -o kt --output-format csv -- python3 $B > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/prof_kt.err
S="--steps 2 --warmup 1 --settle-ms 0 --no-verify"
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
echo "== kernel trace, f16 output"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt_f16 -o kt --output-format csv -- python3 $B --out f16 > $OUT/${TAG}_bench_f16_under_rocprof.json 2> $OUT/prof_kt_f16.err
python3 - <<PY
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob("$OUT/prof_kt_f16/**/*kernel_stats.csv", recursive=True)[0])) if "ggq::" in r["Name"]]
with open("profiles/${TAG}_kernel_stats_f16.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_ns"])
    for r in rows:
        w.writerow([r["Name"].split("(")[0].replace("void ", ""), r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["TotalDurationNs"]])
PY
echo "== kernel trace, quantize / widen / narrow / permute"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt_misc -o kt --output-format csv -- python3 tools/run_misc_kernels.py > $OUT/misc_kernels.json 2> $OUT/prof_kt_misc.err
python3 profiles/summarize_misc.py $OUT/prof_kt_misc $OUT/misc_kernels.json profiles/${TAG}_kernel_stats_misc.json > /dev/null
cp $OUT/${TAG}_bench_under_rocprof.json $OUT/${TAG}_bench_no_prefetch.json $OUT/${TAG}_bench_f16_under_rocprof.json profiles/
mkdir -p $OUT/profiles_out && cp profiles/traffic.json profiles/${TAG}_*.csv profiles/${TAG}_*.json $OUT/profiles_out/
echo done
