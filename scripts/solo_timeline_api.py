# Host API calls next to the kernels / copies of the last `window` ms of a rocprofv3 --hip-trace --kernel-trace --memory-copy-trace run
# (scripts/session_rep3_marks.py: the solo REP3 party is the last proof).  usage: solo_timeline_api.py <dir> <lo ms> <hi ms> [min api ms]
import csv, glob, re, sys
root = sys.argv[1]; lo, hi = float(sys.argv[2]), float(sys.argv[3]); min_api = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"]); name = (m.group(1) if m else r["Kernel_Name"][:40]) + ("<G2>" if "Fp2" in r["Kernel_Name"] else "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", name, "q" + r.get("Queue_Id", "")))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", ""), ""))
kend = max(e[1] for e in ev)
for f in glob.glob(root + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (e - s) / 1e6 >= min_api or r["Function"] in ("hipMemcpyAsync",): ev.append((s, e, "A", r["Function"], "t" + r.get("Thread_Id", "")))
ev.sort()
t0 = kend - int((float(sys.argv[5]) if len(sys.argv) > 5 else 125.0) * 1e6)
for s, e, k, n, q in ev:
    t = (s - t0) / 1e6
    if lo <= t <= hi: print(f"{t:8.2f} +{(e - s) / 1e6:7.3f} {k} {n} {q}")
