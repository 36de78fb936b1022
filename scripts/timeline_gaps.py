"""Busy/idle analysis of one bench step from a rocprofv3 kernel trace: fraction of the step during which NO kernel is running,
and the largest idle gaps with the kernels around them.  usage: python scripts/timeline_gaps.py <kernel_trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
short = lambda k: (re.search(r"(k_\w+)", k).group(1) + ("<G2>" if "Fp2" in k else "")) if "k_" in k else k[:30]
# steps are delimited by k_spmv_csr launches (2 per step): take the span between the 5th and 7th last spmv = one full step
sp = [i for i, e in enumerate(ev) if "k_spmv_csr" in e[2]]
i0, i1 = sp[-6], sp[-4]
seg = ev[i0:i1]
t0, t1 = seg[0][0], ev[i1][0]
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]; gaps = []
for s, e, k in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e - t0, k)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"step span {(t1 - t0) / 1e6:.2f} ms, some kernel running {busy / 1e6:.2f} ms ({100 * busy / (t1 - t0):.1f} %), idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
for g, at, k in sorted(gaps, reverse=True)[:8]:
    print(f"  gap {g / 1e3:8.1f} us at +{at / 1e6:7.2f} ms before {short(k)}")
