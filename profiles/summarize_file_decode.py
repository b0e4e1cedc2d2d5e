"""Kernel-side rate of a file decode from a rocprofv3 --kernel-trace --stats run of
tools/measure_file_decode.py:

    python profiles/summarize_file_decode.py <rocprof dir> <tool's json output> <out.json>

Sums the durations of the decode kernels (deq_one / deq_batch per block type) and divides the
algorithmic bytes of that type (x the number of times the tool decoded it) by them."""
import csv
import glob
import json
import os
import sys


def main():
    d, jpath, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
    info = json.loads(open(jpath).read().strip().splitlines()[-1])
    path = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(path)) if "ggq::deq_" in r["Name"]]
    passes = info["file_decode"]["passes"]
    rec = {"workload": info["workload"], "mode": info["mode"], "file_decode": info["file_decode"], "kernels": []}
    tot_ns, tot_bytes = 0.0, 0.0
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "")
        fmt = name.split("<ggq::")[1].split(",")[0]
        nbytes = info["algorithmic_bytes_by_type"].get(fmt.replace("_", "") if fmt not in info["algorithmic_bytes_by_type"] else fmt)
        k = {"kernel": name, "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"])}
        if nbytes:
            k["algorithmic_GBps"] = round(nbytes * passes / k["total_ns"], 1)
            k["frac_of_8TBps"] = round(k["algorithmic_GBps"] / 8000, 4)
            tot_ns += k["total_ns"]
            tot_bytes += nbytes * passes
        rec["kernels"].append(k)
    if tot_ns:
        rec["kernel_side"] = {"algorithmic_GBps": round(tot_bytes / tot_ns, 1), "frac_of_8TBps": round(tot_bytes / tot_ns / 8000, 4),
                              "launches_per_pass": sum(k["calls"] for k in rec["kernels"]) // passes}
    json.dump(rec, open(out_path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
