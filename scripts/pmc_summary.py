"""Condense rocprofv3 output (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into small files for profiles/.
usage: python scripts/pmc_summary.py <dir holding stats/ pmc_fetch/ pmc_write/>"""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = re.sub(r"cg::", "", name)
    m = re.match(r"(?:void )?(k_\w+)(<.*>)?", name)
    if not m: return name[:60]
    k, t = m.group(1), m.group(2) or ""
    tag = ""
    if any(w in k for w in ("accumulate", "reduce", "window_sum", "merge", "grid", "bitsum", "precompute", "synth", "pack_bases", "check_on_curve")):
        tag = "<G2" if "Fp2" in t else "<G1"
        for a in ("RegAcc29", "LdsAcc29", "RegAcc", "LdsAcc"):
            if a in t: tag += "," + a; break
        tag += ">"
    return k + tag


def counter_pass(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter: continue
                k = short(row["Kernel_Name"])
                acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    return acc


out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline",
       "units": "KiB as reported by rocprofv3. MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced 128-B streams by 2x; the accumulate "
                "kernel issues scattered 64-B gathers (4 x dwordx4 per point), for which the counter is uncalibrated - raw values, no correction applied.",
       "per_launch_avg_KiB": {}}
fetch, write = counter_pass("pmc_fetch", "FETCH_SIZE"), counter_pass("pmc_write", "WRITE_SIZE")
for k in sorted(set(fetch) | set(write)):
    e = {}
    if k in fetch: e["FETCH_SIZE"] = fetch[k][0] / fetch[k][1]; e["launches"] = fetch[k][1]
    if k in write: e["WRITE_SIZE"] = write[k][0] / write[k][1]; e.setdefault("launches", write[k][1])
    if e.get("FETCH_SIZE", 0) + e.get("WRITE_SIZE", 0) > 1024: out["per_launch_avg_KiB"][k] = e
dom = [k for k in out["per_launch_avg_KiB"] if k.startswith("k_msm_accumulate<G1")]
if dom:
    e = out["per_launch_avg_KiB"][dom[0]]
    out["dominant_kernel"] = dom[0]
    out["dominant_kernel_traffic_bytes_per_launch"] = (e.get("FETCH_SIZE", 0) + e.get("WRITE_SIZE", 0)) * 1024
json.dump(out, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)

# kernel stats: keep the rocprofv3 summary as it is, with shortened names, top 25 rows
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(root, "kernel_stats.csv"), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:25]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    for r in rows[:8]: print(short(r["Name"]), r["Calls"], r["AverageNs"], r["Percentage"])
print(json.dumps(out.get("per_launch_avg_KiB", {}), indent=1)[:1500])
