"""Kernel / copy timeline of the LAST `window_ms` of a rocprofv3 --kernel-trace [--memory-copy-trace] run (the last proof of a session leg):
one line per event with its start offset, duration and queue.  usage: proof_timeline.py <dir> <window_ms> [segments]
segments: the run is cut where the device was idle for > 30 ms (between the legs of scripts/later_leg.py) and the last window of EVERY piece that ran
the small-circuit row kernel is printed."""
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
def show(ev):
    end = max(e[1] for e in ev if e[3] != "k_probe_spin")
    t0 = end - int(window * 1e6)
    for s, e, k, n, q, g in ev:
        if s >= t0 and e <= end:
            print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {k} q{q:>3} grid {g:>8}  {n}")
if len(sys.argv) > 3 and sys.argv[3] == "segments":
    pieces, cur = [], [ev[0]]
    for x in ev[1:]:
        if x[0] - max(y[1] for y in cur[-8:]) > 30e6: pieces.append(cur); cur = []
        cur.append(x)
    pieces.append(cur)
    for i, pc in enumerate(pieces):
        if sum(1 for x in pc if x[3] == "k_spmv_csr_wave") >= 8 and not any(x[3] == "k_spmv_csr" for x in pc):
            print(f"== piece {i}: {len(pc)} events over {(pc[-1][1] - pc[0][0]) / 1e6:.1f} ms"); show(pc)
else:
    show(ev)
