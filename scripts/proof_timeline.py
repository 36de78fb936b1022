"""Kernel / copy timeline of the LAST `window_ms` of a rocprofv3 --kernel-trace [--memory-copy-trace] run (the last proof of a session leg):
one line per event with its start offset, duration and queue.  usage: proof_timeline.py <dir> <window_ms>"""
import csv, glob, re, sys
root, window = sys.argv[1], float(sys.argv[2])
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"]); name = (m.group(1) if m else r["Kernel_Name"][:40]) + ("<G2>" if "Fp2" in r["Kernel_Name"] else "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", name, r.get("Queue_Id", ""), r.get("Grid_Size_X", r.get("Grid_Size", ""))))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "copy")), "", ""))
ev.sort()
end = max(e[1] for e in ev)
t0 = end - int(window * 1e6)
for s, e, k, n, q, g in ev:
    if s >= t0:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {k} q{q:>3} grid {g:>8}  {n}")
