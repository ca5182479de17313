"""Aggregate rocprofv3 --pmc passes into per-kernel summaries.

    python profiles/summarize_pmc.py traffic <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
    python profiles/summarize_pmc.py mfma <counter_collection.csv> <out.json>

traffic: HBM bytes per launch.  Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the
counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming
read, so the read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
mfma: matrix-core utilisation per launch = SQ_VALU_MFMA_BUSY_CYCLES / (active cycles x 1,024 SIMDs), active cycles =
GRBM_GUI_ACTIVE / 8 (rocprofv3 reports the sum over the 8 XCDs, same guide, DVFS section); and the MFMA flops the
hardware counted (SQ_INSTS_VALU_MFMA_MOPS_* x 512).
Only the largest launches of each kernel (the benchmark-sized ones) are averaged."""
import csv
import json
import os
import sys
from collections import defaultdict

PREFIXES = ("wn::", "void wn::", "w16::", "void w16::")


def load(path, counter):
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and r["Kernel_Name"].startswith(PREFIXES):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                rows[name].append((int(r["Grid_Size"]), float(r["Counter_Value"]),
                                   int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Dispatch_Id", "")))
    return rows


def biggest(rows):
    gmax = max(g for g, _, _, _ in rows)
    return [(v, ns) for g, v, ns, _ in rows if g == gmax]


def traffic(fetch_csv, write_csv, out):
    fetch, write = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    for name in sorted(set(fetch) | set(write)):
        ent = {}
        for key, rows, corr in (("fetch", fetch.get(name, []), 2.0), ("write", write.get(name, []), 1.0)):
            if not rows:
                continue
            big = biggest(rows)
            ent[key + "_bytes_per_launch"] = corr * 1024.0 * sum(v for v, _ in big) / len(big)
            ent[key + "_launches_averaged"] = len(big)
            ent["avg_ns_under_pmc"] = sum(ns for _, ns in big) / len(big)
        ent["hbm_bytes_per_launch"] = ent.get("fetch_bytes_per_launch", 0.0) + ent.get("write_bytes_per_launch", 0.0)
        res[name] = ent
    json.dump({"note": "FETCH_SIZE x2 (gfx950 correction), KiB -> bytes; largest-grid launches only",
               "commit": os.environ.get("WN_PROFILE_COMMIT"), "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-56s %10.1f MB read %10.1f MB written" % (k[:56], v.get("fetch_bytes_per_launch", 0) / 1e6,
                                                         v.get("write_bytes_per_launch", 0) / 1e6))


def mfma(pmc_csv, out):
    busy, act = load(pmc_csv, "SQ_VALU_MFMA_BUSY_CYCLES"), load(pmc_csv, "GRBM_GUI_ACTIVE")
    bf16, f32 = load(pmc_csv, "SQ_INSTS_VALU_MFMA_MOPS_BF16"), load(pmc_csv, "SQ_INSTS_VALU_MFMA_MOPS_F32")
    f16 = load(pmc_csv, "SQ_INSTS_VALU_MFMA_MOPS_F16")
    res = {}
    for name in sorted(busy):
        b, a = biggest(busy[name]), biggest(act.get(name, []) or busy[name])
        n = min(len(b), len(a))
        if n == 0:
            continue
        cyc = sum(v for v, _ in a[:n]) / n / 8.0                      # active cycles of the launch (sum over 8 XCDs / 8)
        bsy = sum(v for v, _ in b[:n]) / n
        ent = {"launches_averaged": n, "mfma_busy_cycles": bsy, "active_cycles": cyc,
               "mfma_util": bsy / (cyc * 1024.0) if cyc > 0 else None, "avg_ns_under_pmc": sum(ns for _, ns in b[:n]) / n}
        for key, rows in (("bf16", bf16.get(name)), ("f16", f16.get(name)), ("f32", f32.get(name))):
            if rows:
                bb = biggest(rows)
                ent["mfma_flops_%s" % key] = 512.0 * sum(v for v, _ in bb) / len(bb)
        res[name] = ent
    json.dump({"note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); flops = MOPS x 512",
               "commit": os.environ.get("WN_PROFILE_COMMIT"), "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["avg_ns_under_pmc"]):
        print("%-56s util %5.1f %%   %8.1f us   %.3g bf16 flop  %.3g f16 flop  %.3g f32 flop" % (
            k[:56], 100 * (v["mfma_util"] or 0), v["avg_ns_under_pmc"] / 1e3, v.get("mfma_flops_bf16", 0),
            v.get("mfma_flops_f16", 0), v.get("mfma_flops_f32", 0)))


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(*sys.argv[2:5])
    elif sys.argv[1] == "mfma":
        mfma(*sys.argv[2:4])
    else:                                # round-1 calling convention
        traffic(*sys.argv[1:4])
