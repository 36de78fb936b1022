import csv, glob, sys, re
root = sys.argv[1]
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"]); name = (m.group(1) if m else r["Kernel_Name"][:30]) + ("<G2>" if "Fp2" in r["Kernel_Name"] else "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", name, r.get("Queue_Id", "")))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "copy")), ""))
ev.sort()
end = ev[-1][1]
t0 = end - int(float(sys.argv[2]) * 1e6)
last = [e for e in ev if e[0] >= t0]
# coarse listing: merge consecutive same-name events
out = []
for s, e, k, n, q in last:
    if out and out[-1][3] == n and out[-1][2] == k and s - out[-1][1] < 300000: out[-1][1] = max(out[-1][1], e); out[-1][4] += 1
    else: out.append([s, e, k, n, 1])
for s, e, k, n, c in out:
    if (e - s) > 150000 or k == "C" and (e - s) > 500000 or "rep3" in n or "spmv" in n or "vec_binary" in n:
        print(f"{(s - t0) / 1e6:8.2f} ms  +{(e - s) / 1e6:7.2f}  {k} {n} x{c}")
