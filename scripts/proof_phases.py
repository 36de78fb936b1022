# Per-proof phase offsets of the last proofs of a rocprofv3 --kernel-trace --memory-copy-trace run of scripts/session_leg.py (one REP3 party per
# proof): where each proof's draws, aux accumulations, witness map and quotient MSM started and ended, relative to the proof's first share upload.
# usage: proof_phases.py <trace dir> <proofs>
import csv, glob, re, sys
root, want = sys.argv[1], int(sys.argv[2])
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"]); name = (m.group(1) if m else r["Kernel_Name"][:30]) + ("<G2>" if "Fp2" in r["Kernel_Name"] else "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
spmv = [e[0] for e in ev if e[2] == "k_spmv_csr"]
starts = [spmv[i] for i in range(len(spmv)) if i == 0 or spmv[i] - spmv[i - 1] > 1_000_000]      # first SpMV of each proof
starts = starts[-want:]
for pi, s0 in enumerate(starts):
    lo = s0 - 3_000_000; hi = starts[pi + 1] - 3_000_000 if pi + 1 < len(starts) else ev[-1][1] + 1
    ch = [e for e in ev if lo <= e[0] < hi and e[2] == "k_chacha_candidates"]
    if not ch: continue
    t0 = ch[-2][0] if len(ch) >= 2 else ch[0][0]                                                     # the proof's first draw
    win = [e for e in ev if t0 <= e[0] < t0 + 8_000_000 and e[0] < hi + 3_000_000]
    def first(n, k=0): l = [e for e in win if e[2] == n]; return (l[k][0] - t0) / 1e6 if len(l) > k else float("nan")
    def lastend(n): l = [e for e in win if e[2] == n]; return (l[-1][1] - t0) / 1e6 if l else float("nan")
    accs = [e for e in win if e[2].startswith("k_msm_accumulate_pf")]
    print(f"proof {pi}: g2acc {first('k_msm_accumulate_pf<G2>'):.2f} g1acc {first('k_msm_accumulate_pf'):.2f}-{(accs[1][1]-t0)/1e6 if len(accs)>1 else 0:.2f} spmv {first('k_spmv_csr'):.2f} "
          f"mul1 {first('k_rep3_mul_local'):.2f} mul2 {first('k_rep3_mul_local',1):.2f} aux_last_bitsum_final {max((e[1]-t0)/1e6 for e in win if e[2].startswith('k_msm_bitsum_final')) :.2f} "
          f"h_acc {(accs[-1][0]-t0)/1e6:.2f}-{(accs[-1][1]-t0)/1e6:.2f} last {max((e[1]-t0)/1e6 for e in win if e[2].startswith('k_msm_bitsum_final')):.2f}")
