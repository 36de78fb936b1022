import csv, glob, sys, re
root = sys.argv[1]; lo, hi = float(sys.argv[2]), float(sys.argv[3])
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"]); name = (m.group(1) if m else r["Kernel_Name"][:40]) + ("<G2>" if "Fp2" in r["Kernel_Name"] else "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", name + " grid=" + r.get("Grid_Size", "?"), r.get("Queue_Id", "")))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    if rows: print("copy trace columns:", list(rows[0].keys()))
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "?")), ""))
ev.sort()
end = ev[-1][1]; t0 = end - int((float(sys.argv[4]) if len(sys.argv) > 4 else 125.0) * 1e6)   # origin: that many ms before the last event
for s, e, k, n, q in ev:
    t = (s - t0) / 1e6
    if lo <= t <= hi: print(f"{t:8.2f} +{(e - s) / 1e6:7.3f} {k} {n} q={q}")
