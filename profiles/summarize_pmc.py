"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

Usage: python profiles/summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so the
read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Only the largest launches
of each kernel (the config-2 sized ones) are averaged."""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and r["Kernel_Name"].startswith(("wn::", "void wn::")):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                rows[name].append((int(r["Grid_Size"]), float(r["Counter_Value"]),
                                   int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return rows


def main(fetch_csv, write_csv, out):
    fetch, write = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    for name in sorted(set(fetch) | set(write)):
        ent = {}
        for key, rows, corr in (("fetch", fetch.get(name, []), 2.0), ("write", write.get(name, []), 1.0)):
            if not rows:
                continue
            gmax = max(g for g, _, _ in rows)
            big = [(v, ns) for g, v, ns in rows if g == gmax]
            ent[key + "_bytes_per_launch"] = corr * 1024.0 * sum(v for v, _ in big) / len(big)
            ent[key + "_launches_averaged"] = len(big)
            ent["avg_ns_under_pmc"] = sum(ns for _, ns in big) / len(big)
        ent["hbm_bytes_per_launch"] = ent.get("fetch_bytes_per_launch", 0.0) + ent.get("write_bytes_per_launch", 0.0)
        res[name] = ent
    json.dump({"note": "FETCH_SIZE x2 (gfx950 correction), KiB -> bytes; largest-grid launches only", "kernels": res},
              open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-40s %10.1f MB read %10.1f MB written" % (k, v.get("fetch_bytes_per_launch", 0) / 1e6,
                                                         v.get("write_bytes_per_launch", 0) / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
