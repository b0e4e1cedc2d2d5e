"""python profiles/summarize_misc.py <rocprof dir> <json printed by tools/run_misc_kernels.py> <out.json>
Launch durations of each kernel from a rocprofv3 --kernel-trace run (average, median and minimum over the
per-dispatch trace; and the same over the second half of a kernel's launches only: a kernel's first ~20 ms after a
change of load run up to 20 % slow, measured, so a run of 30 one-millisecond launches that starts right behind
another kernel is mostly that transient), priced in algorithmic bytes against 8 TB/s."""
import csv
import glob
import json
import os
import statistics
import sys

csv.field_size_limit(1 << 30)
d, jpath, out = sys.argv[1:4]
want = json.loads(open(jpath).read().strip().splitlines()[-1])
trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
durs = {}
for r in csv.DictReader(open(trace)):
    if "ggq::" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    durs.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for name, ds in durs.items():
    late = ds[len(ds) // 2:]
    rec = {"kernel": name, "calls": len(ds), "avg_ns": round(sum(ds) / len(ds), 1),
           "median_ns": statistics.median(ds), "min_ns": min(ds), "second_half_avg_ns": round(sum(late) / len(late), 1)}
    for key, nbytes in want.items():
        if key in name:
            rec["algorithmic_bytes_per_launch"] = nbytes
            rec["algorithmic_GBps_avg"] = round(nbytes / rec["avg_ns"], 1)
            rec["algorithmic_GBps_median"] = round(nbytes / rec["median_ns"], 1)
            rec["frac_of_8TBps_avg"] = round(nbytes / rec["avg_ns"] / 8000, 4)
            rec["frac_of_8TBps_median"] = round(nbytes / rec["median_ns"] / 8000, 4)
            rec["frac_of_8TBps_second_half_avg"] = round(nbytes / rec["second_half_avg_ns"] / 8000, 4)
    rows.append(rec)
json.dump(rows, open(out, "w"), indent=1)
for r in rows:
    print(r["kernel"][:64], r["calls"], r.get("frac_of_8TBps_avg"), r.get("frac_of_8TBps_median"), r.get("frac_of_8TBps_second_half_avg"))
