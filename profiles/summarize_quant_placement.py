"""Joins tools/probe_quant_placement.py's own record with the rocprofv3 runs of the same script:
    python profiles/summarize_quant_placement.py plain.json kt_dir fetch_dir write_dir out.json
Per placement: fraction of 8 TB/s from HIP events (unprofiled run) and from the kernel trace (median, min, max
over the 30 launches), FETCH_SIZE x 2 and WRITE_SIZE per launch (KiB counters, FETCH doubled as for every wide
read stream on gfx950) against the algorithmic bytes."""
import csv
import glob
import json
import os
import statistics
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def main():
    plain, kt, fetch, write, out = sys.argv[1:6]
    doc = json.loads(open(plain).read().strip().splitlines()[-1])
    reps, alg = doc["launches_per_placement"], doc["algorithmic_bytes"]
    durs = []
    for r in csv.DictReader(open(find(kt, "kernel_trace.csv"))):
        if "quant_one<ggq::EncQ4_0, ggq::InF32" in r["Kernel_Name"]:
            durs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    durs = [d for _, d in sorted(durs)]

    def counter(d, name):
        vals = []
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            if r["Counter_Name"] == name and "quant_one<ggq::EncQ4_0, ggq::InF32" in r["Kernel_Name"]:
                vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        return [v for _, v in sorted(vals)]

    fe, wr = counter(fetch, "FETCH_SIZE"), counter(write, "WRITE_SIZE")
    for k, p in enumerate(doc["placements"]):
        p.pop("ms", None)
        d = durs[k * reps:(k + 1) * reps]
        if d:
            fr = [alg / (x * 1e-9) / 8e12 for x in d]
            p["kernel_trace"] = {"frac_median": round(statistics.median(fr), 4), "frac_min": round(min(fr), 4),
                                 "frac_max": round(max(fr), 4), "launches": len(d)}
        f, w = fe[k * reps:(k + 1) * reps], wr[k * reps:(k + 1) * reps]
        if f and w:
            rb, wb = 2048.0 * statistics.mean(f), 1024.0 * statistics.mean(w)
            p["counters"] = {"read_bytes_FETCH_SIZE_x2": round(rb), "write_bytes_WRITE_SIZE": round(wb),
                             "traffic_over_algorithmic": round((rb + wb) / alg, 4),
                             "read_spread": round((max(f) - min(f)) / statistics.mean(f), 4)}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


main()
