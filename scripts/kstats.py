"""Average duration per kernel from a rocprofv3 --kernel-trace run.  usage: kstats.py <dir> [substring ...]"""
import csv, glob, re, sys, collections
root, pats = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"]); name = (m.group(1) if m else r["Kernel_Name"][:50]) + ("<G2>" if "Fp2" in r["Kernel_Name"] else "")
        acc[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if pats and not any(p in name for p in pats): continue
    v2 = sorted(v)
    print(f"{name:40s} n={len(v):4d} avg={sum(v)/len(v):9.1f} us  min={v2[0]:9.1f}  med={v2[len(v2)//2]:9.1f}  total={sum(v)/1e3:8.2f} ms")
