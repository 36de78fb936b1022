"""Where a bench step spends its time: from a rocprofv3 --kernel-trace run of bench.py, the last step's kernels — time covered by an
accumulate launch, time covered by nothing, and the kernels that run while no accumulate does.  usage: step_timeline.py <dir> [step_ms=75]"""
import csv, glob, re, sys
root = sys.argv[1]; step_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 75.0
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"]); name = (m.group(1) if m else r["Kernel_Name"][:40]) + ("<G2>" if "Fp2" in r["Kernel_Name"] else "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "")))
ev.sort()
sp = [e for e in ev if "spmv" in e[2]]                       # every step starts with two constraint evaluations; the isolated launches after the steps have none
lo = sp[-2][0] - 200_000; end = lo + int(step_ms * 1e6)
win = [e for e in ev if e[1] > lo and e[0] < end]
def union(iv):
    iv = sorted(iv); out = []
    for s, e in iv:
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
ua = union([(max(s, lo), min(e, end)) for s, e, n, q in win if "accumulate" in n])
uall = union([(max(s, lo), min(e, end)) for s, e, n, q in win])
cov = lambda u: sum(e - s for s, e in u) / 1e6
print(f"window {step_ms} ms: accumulate running {cov(ua):.2f} ms, any kernel running {cov(uall):.2f} ms, idle {step_ms - cov(uall):.2f} ms")
# gaps between accumulate coverage: what runs there
gaps = []; prev = lo
for s, e in ua:
    if s - prev > 50e3: gaps.append((prev, s))
    prev = e
if end - prev > 50e3: gaps.append((prev, end))
for gs, ge in gaps:
    inside = {}
    for s, e, n, q in win:
        o = min(e, ge) - max(s, gs)
        if o > 0: inside[n] = inside.get(n, 0) + o / 1e3
    top = sorted(inside.items(), key=lambda kv: -kv[1])[:6]
    print(f"  gap at {(gs - lo) / 1e6:7.2f} ms, {(ge - gs) / 1e6:6.2f} ms: " + ", ".join(f"{n} {t:.0f}us" for n, t in top))
print("accumulate launches in the window:")
for s, e, n, q in win:
    if "accumulate" in n: print(f"  {(s - lo) / 1e6:7.2f} +{(e - s) / 1e6:6.2f} {n} q{q}")
if len(sys.argv) > 3:
    thr = float(sys.argv[3])
    print(f"kernels >= {thr} ms in the window:")
    for s, e, n, q in win:
        if (e - s) / 1e6 >= thr: print(f"  {(s - lo) / 1e6:7.2f} +{(e - s) / 1e6:6.2f} {n} q{q}")
