"""python profiles/summarize_misc.py <rocprof dir> <json printed by tools/run_misc_kernels.py> <out.json>
Average launch duration of each kernel from rocprofv3 --kernel-trace --stats, priced in algorithmic bytes."""
import csv
import glob
import json
import os
import sys

d, jpath, out = sys.argv[1:4]
want = json.loads(open(jpath).read().strip().splitlines()[-1])
path = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(path)):
    if "ggq::" not in r["Name"]:
        continue
    name = r["Name"].split("(")[0].replace("void ", "")
    rec = {"kernel": name, "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"])}
    for key, nbytes in want.items():
        if key in name:
            rec["algorithmic_bytes_per_launch"] = nbytes
            rec["algorithmic_GBps"] = round(nbytes / rec["avg_ns"], 1)
            rec["frac_of_8TBps"] = round(nbytes / rec["avg_ns"] / 8000, 4)
    rows.append(rec)
json.dump(rows, open(out, "w"), indent=1)
print(json.dumps(rows, indent=1))
