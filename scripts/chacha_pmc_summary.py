"""Condense the rocprofv3 passes over scripts/chacha_timing.py (kernel stats, FETCH_SIZE, WRITE_SIZE) into profiles/r03_chacha_*.
usage: python scripts/chacha_pmc_summary.py <dir holding stats/ pmc_fetch/ pmc_write/>
Counter values are KiB; stream factors from profiles/r03_pmc_calibration.json (reads of wide streams are counted half, writes in full)."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
READ_F, WRITE_F = 1.9999, 1.0


def short(n): return n.split("(")[0].replace("cg::", "")


def trace(d):
    rows = []
    for f in glob.glob(os.path.join(root, d, "**", "*kernel_trace.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "chacha" in r["Kernel_Name"]]
    return rows


def counters(d, name):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != name or "chacha" not in r["Kernel_Name"]: continue
            k = (short(r["Kernel_Name"]), int(r.get("Grid_Size") or r.get("Grid_Size_X")))
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


dur = defaultdict(list)
for r in trace("stats"):
    g = int(r.get("Grid_Size_X") or r.get("Grid_Size"))
    dur[(short(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
fetch, write = counters("pmc_fetch", "FETCH_SIZE"), counters("pmc_write", "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (three separate runs) -- python scripts/chacha_timing.py",
       "note": "lanes = stream blocks = candidate pairs of one call; algorithmic bytes: candidates 64 B written per lane, compaction 64 B read per lane + 32 B written per "
               "accepted draw; traffic = FETCH_SIZE x 1024 x 2.0 (wide reads are counted half, r03_pmc_calibration.json) + WRITE_SIZE x 1024",
       "kernels": []}
for k in sorted(set(dur) | set(fetch) | set(write), key=lambda x: (x[0], x[1])):
    name, lanes = k
    e = {"kernel": name, "lanes": lanes}
    if k in dur: e["launches"] = len(dur[k]); e["avg_us"] = sum(dur[k]) / len(dur[k]) / 1e3
    if k in fetch: e["FETCH_SIZE_KiB"] = fetch[k][0] / fetch[k][1]
    if k in write: e["WRITE_SIZE_KiB"] = write[k][0] / write[k][1]
    if "FETCH_SIZE_KiB" in e and "WRITE_SIZE_KiB" in e:
        e["traffic_bytes"] = e["FETCH_SIZE_KiB"] * 1024 * READ_F + e["WRITE_SIZE_KiB"] * 1024 * WRITE_F
        if "avg_us" in e: e["traffic_GBs"] = e["traffic_bytes"] / (e["avg_us"] * 1e-6) / 1e9
    if name == "k_chacha_candidates": e["algorithmic_bytes"] = 64 * lanes
    if name == "k_chacha_compact": e["algorithmic_bytes_upper"] = 64 * lanes + 64 * lanes          # every candidate accepted
    if "avg_us" in e and "algorithmic_bytes" in e: e["algorithmic_GBs"] = e["algorithmic_bytes"] / (e["avg_us"] * 1e-6) / 1e9
    out["kernels"].append(e)
json.dump(out, open(os.path.join(root, "chacha_pmc.json"), "w"), indent=1)
for e in out["kernels"]:
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items()})
