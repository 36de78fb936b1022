"""Host-side HIP API timeline of the LAST `window_ms` of a rocprofv3 --hip-trace run: the calls that took longest and the totals per function
(what the host spends between two kernels of a small proof).  usage: hip_api_timeline.py <dir> <window_ms> [min_us=15]"""
import collections, csv, glob, sys
root, window = sys.argv[1], float(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
rows = []
for f in glob.glob(root + "/**/*hip_api_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
# anchor: the end of the last KERNEL (the last proof's last reduction), not the end of the process (its tear-down frees everything)
kend = [int(r["End_Timestamp"]) for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f)) if "k_probe_spin" not in r["Kernel_Name"]]     # (contexts made at the very end of a process probe their streams)
end = (max(kend) + 400000) if kend else max(int(r["End_Timestamp"]) for r in rows); t0 = end - int(window * 1e6)
rows = [r for r in rows if int(r["Start_Timestamp"]) <= end]
tot = collections.defaultdict(lambda: [0, 0.0])
print(f"calls of at least {min_us} us in the last {window} ms:")
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0: continue
    tot[r["Function"]][0] += 1; tot[r["Function"]][1] += (e - s) / 1e3
    if (e - s) / 1e3 >= min_us:
        print(f"{(s - t0) / 1e3:9.1f} us +{(e - s) / 1e3:8.1f}  tid {r.get('Thread_Id', '')}  {r['Function']}")
print("totals per function:")
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{us:9.1f} us  {n:5d} calls  {k}")
