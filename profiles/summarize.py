"""Condenses rocprofv3 output (gpurun_out/prof_*) into the small files kept under profiles/.

    python profiles/summarize.py r01 gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write \
        llama7b_q4_k_m:f32

Writes profiles/<tag>_kernel_stats.csv (rows of this library's kernels only; the synthetic-input
kernels of torch are dropped) and merges HBM traffic per launch into profiles/traffic.json.
Counter handling on gfx950: FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports half the
bytes of a wide coalesced read stream (128-byte requests tallied as 64), so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.  The two counters come from separate runs.
"""
import csv
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    tag, kt, fetch, write, key = sys.argv[1:6]
    rows = []
    path = find(kt, "kernel_stats.csv")
    if path:
        for r in csv.DictReader(open(path)):
            if "ggq::" in r["Name"]:
                rows.append({"kernel": short(r["Name"]), "calls": r["Calls"], "avg_ns": r["AverageNs"],
                             "min_ns": r["MinNs"], "max_ns": r["MaxNs"], "total_ns": r["TotalDurationNs"]})
    else:   # rocprofv3's default output: one rocpd SQLite database
        import sqlite3
        db = sqlite3.connect(find(kt, "_results.db"))
        for name, n, avg, lo, hi, tot in db.execute(
                "select name, count(*), avg(duration), min(duration), max(duration), sum(duration) "
                "from kernels where name like '%ggq::%' group by name order by sum(duration) desc"):
            rows.append({"kernel": short(name), "calls": n, "avg_ns": round(avg), "min_ns": lo, "max_ns": hi,
                         "total_ns": tot})
    with open(os.path.join(HERE, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)

    def counter(d, cname):
        agg = {}
        path = find(d, "counter_collection.csv")
        if path:
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] == cname and "ggq::" in r["Kernel_Name"]:
                    agg.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
        else:
            import sqlite3
            db = sqlite3.connect(find(d, "_results.db"))
            for name, value in db.execute("select kernel_name, value from counters_collection "
                                          "where counter_name = ? and kernel_name like '%ggq::%'", (cname,)):
                agg.setdefault(short(name), []).append(float(value))
        return {k: sum(v) / len(v) for k, v in agg.items()}

    fe, wr = counter(fetch, "FETCH_SIZE"), counter(write, "WRITE_SIZE")
    tpath = os.path.join(HERE, "traffic.json")
    doc = json.load(open(tpath)) if os.path.exists(tpath) else {}
    written = set()
    for k in fe:
        if "deq_batch<ggq::" not in k and "deq_one<ggq::" not in k and "quant_one<ggq::" not in k:
            continue        # helper kernels (the tile table builder) carry no format
        fmt = k.split("<ggq::")[1].split(",")[0]
        read_b = 2.0 * fe[k] * 1024.0
        write_b = wr.get(k, 0.0) * 1024.0
        # a plan can hold a table launch and single-tensor launches of one format: the entry is the larger launch
        old = doc.get(f"{key}:{fmt}") if f"{key}:{fmt}" in written else None
        if old and old["hbm_bytes_per_launch"] > read_b + write_b:
            continue
        written.add(f"{key}:{fmt}")
        doc[f"{key}:{fmt}"] = {
            "kernel": k, "round": tag,
            "FETCH_SIZE_KiB_raw": fe[k], "WRITE_SIZE_KiB_raw": wr.get(k),
            "read_bytes_corrected_x2": read_b, "write_bytes": write_b,
            "hbm_bytes_per_launch": read_b + write_b,
        }
    json.dump(doc, open(tpath, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
