"""Condense the rocprofv3 output of scripts/collect_profiles_r06.sh into the small files kept under profiles/.
usage: python scripts/profile_summary_r06.py gpurun_out/r06"""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = re.sub(r"cg::", "", name)
    m = re.match(r"(?:void )?(k_\w+)(<.*>)?", name)
    if not m:
        return name[:60]
    k, t = m.group(1), m.group(2) or ""
    tag = ""
    if any(w in k for w in ("accumulate", "reduce", "window_sum", "merge", "grid", "bitsum", "precompute", "synth", "pack_bases", "check_on_curve", "bitsum")):
        tag = "<G2>" if "Fp2" in t else "<G1>"
    return k + tag


def counters(d):
    """{kernel: {counter: (sum, launches)}} plus per-dispatch durations from the kernel trace of the same run"""
    dur = {}
    for f in glob.glob(os.path.join(root, d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0, 0]))
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            e = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            e[0] += float(r["Counter_Value"]); e[1] += 1; e[2] += dur.get(r["Dispatch_Id"], 0)
    return acc


def stats_csv(d, out_name, header):
    for f in glob.glob(os.path.join(root, d, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(root, out_name), "w") as fh:
            for line in header:
                fh.write("# " + line + "\n")
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows[:32]:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
        print(out_name); [print("  ", short(r["Name"]), r["Calls"], r["AverageNs"]) for r in rows[:10]]


stats_csv("stats", "r06_kernel_stats.csv", ["rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-session   (MI355X, 2^22 BN254 REP3 step)",
                                              "timed steps: four streams on two contexts run concurrently, kernel durations include CU sharing; the run also holds the isolated launches of bench.py's roofline section"])
stats_csv("serial", "r06_serial_kernel_stats.csv", ["rocprofv3 --kernel-trace --stats -- python scripts/serial_kernels.py 22 5   (every kernel of the step ALONE on the GPU, one share component per MSM call)",
                                                     "these per-launch averages are what bench.py reports as isolated_ms / roofline.launch_ms (HIP events on the kernels' own stream)"])

stats_csv("serial_bls", "r06_serial_kernel_stats_bls12_381.csv", ["rocprofv3 --kernel-trace --stats -- python scripts/serial_kernels.py 22 3 bls12_381   (BLS12-381: every kernel of the 2^22 step ALONE on the GPU; 96 / 192-byte points, 14 x 28-bit lazy limbs in Fq)"])
stats_csv("session_poseidon", "r06_session_poseidon_kernel_stats.csv", ["rocprofv3 --kernel-trace --stats -- python scripts/session_leg.py poseidon 20   (NO_EXTRAS=1: one REP3 party of the reference's own bench circuit, the Poseidon fixture m = 256, through cgh_session_prove_rep3_party_ex)"])
stats_csv("session", "r06_session_kernel_stats.csv", ["rocprofv3 --kernel-trace --stats -- python scripts/session_leg.py 22 5   (NO_EXTRAS=1: one REP3 party through cgh_session_prove_rep3_party_ex, its ChaCha12 draws inside the call; three-party warm-up + record run included)"])

# calibration of FETCH_SIZE / WRITE_SIZE on known byte counts
known = {"k_cal_stream": ((4 << 30), (4 << 30)), "k_cal_gather": None}
cal = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- scripts/_build/pmc_calibrate", "kernels": {}}
cf, cw = counters("cal_fetch"), counters("cal_write")
sizes = {"k_cal_stream": {"read": 4 << 30, "write": 4 << 30}}
for k in set(cf) | set(cw):
    if not k.startswith("k_cal_") or "fill" in k:
        continue
    e = {}
    if "FETCH_SIZE" in cf.get(k, {}): e["FETCH_SIZE_KiB_per_launch"] = cf[k]["FETCH_SIZE"][0] / cf[k]["FETCH_SIZE"][1]
    if "WRITE_SIZE" in cw.get(k, {}): e["WRITE_SIZE_KiB_per_launch"] = cw[k]["WRITE_SIZE"][0] / cw[k]["WRITE_SIZE"][1]
    cal["kernels"][k] = e
# the two gather instantiations share a short name; split them by the launch order recorded in the raw files
def per_launch(d, counter, kernel_sub):
    vals = []
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return vals
g4f, g8f = per_launch("cal_fetch", "FETCH_SIZE", "k_cal_gather<4>") or per_launch("cal_fetch", "FETCH_SIZE", "Li4E"), per_launch("cal_fetch", "FETCH_SIZE", "k_cal_gather<8>") or per_launch("cal_fetch", "FETCH_SIZE", "Li8E")
g4w, g8w = per_launch("cal_write", "WRITE_SIZE", "k_cal_gather<4>") or per_launch("cal_write", "WRITE_SIZE", "Li4E"), per_launch("cal_write", "WRITE_SIZE", "k_cal_gather<8>") or per_launch("cal_write", "WRITE_SIZE", "Li8E")
avg = lambda v: sum(v) / len(v) if v else None
factors = {}
st = cal["kernels"].get("k_cal_stream", {})
if st.get("FETCH_SIZE_KiB_per_launch"): factors["stream_read"] = (4 << 30) / (st["FETCH_SIZE_KiB_per_launch"] * 1024)
if st.get("WRITE_SIZE_KiB_per_launch"): factors["stream_write"] = (4 << 30) / (st["WRITE_SIZE_KiB_per_launch"] * 1024)
if avg(g4f): factors["gather64_read"] = ((1 << 26) * 64 + (1 << 26) * 4) / (avg(g4f) * 1024)
if avg(g8f): factors["gather128_read"] = ((1 << 25) * 128 + (1 << 25) * 4) / (avg(g8f) * 1024)
if avg(g4w): factors["dword_write"] = ((1 << 26) * 4) / (avg(g4w) * 1024)
cal["true_bytes_over_counter_bytes"] = factors
cal["reading"] = "factor = known bytes / (counter x 1024); applied to the traffic figures of profiles/r06_pmc_traffic.json (gather factor for the bucket accumulation's reads, stream factors elsewhere)"
json.dump(cal, open(os.path.join(root, "r06_pmc_calibration.json"), "w"), indent=1)
print("calibration", json.dumps(factors))

# traffic per launch of the serial kernels
fetch, write = counters("pmc_fetch"), counters("pmc_write")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python scripts/serial_kernels.py 22 2",
       "units": "counter KiB per launch as rocprofv3 reports them, and bytes after the calibration of r06_pmc_calibration.json (FETCH_SIZE tallies 128-byte "
                "requests at 64 bytes for wide streams; the factor for the accumulation's scattered 64 / 128-byte gathers is measured there)",
       "per_launch": {}}
for k in sorted(set(fetch) | set(write)):
    e = {}
    if "FETCH_SIZE" in fetch.get(k, {}): e["FETCH_SIZE_KiB"] = fetch[k]["FETCH_SIZE"][0] / fetch[k]["FETCH_SIZE"][1]; e["launches"] = fetch[k]["FETCH_SIZE"][1]
    if "WRITE_SIZE" in write.get(k, {}): e["WRITE_SIZE_KiB"] = write[k]["WRITE_SIZE"][0] / write[k]["WRITE_SIZE"][1]
    if e.get("FETCH_SIZE_KiB", 0) + e.get("WRITE_SIZE_KiB", 0) < 1024:
        continue
    gather = "accumulate" in k
    rf = factors.get("gather128_read" if "<G2>" in k else "gather64_read", 1.0) if gather else factors.get("stream_read", 2.0)
    wf = factors.get("stream_write", 1.0)
    e["read_bytes_calibrated"] = e.get("FETCH_SIZE_KiB", 0) * 1024 * rf
    e["write_bytes_calibrated"] = e.get("WRITE_SIZE_KiB", 0) * 1024 * wf
    out["per_launch"][k] = e
dom = [k for k in out["per_launch"] if k.startswith("k_msm_accumulate") and "<G1>" in k]
if dom:
    e = out["per_launch"][dom[0]]
    out["dominant_kernel"] = dom[0]
    out["dominant_kernel_traffic_bytes_per_launch"] = e["read_bytes_calibrated"] + e["write_bytes_calibrated"]
    out["dominant_kernel_algorithmic_bytes_per_launch"] = 96 * ((1 << 22))
json.dump(out, open(os.path.join(root, "r06_pmc_traffic.json"), "w"), indent=1)
print("traffic", json.dumps({k: (round(v["read_bytes_calibrated"] / 1e6), round(v["write_bytes_calibrated"] / 1e6)) for k, v in out["per_launch"].items()}))

# SQ counters and clocks
sq, clk = counters("pmc_sq"), counters("pmc_clk")
res = {"source": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES "
                 "and --pmc GRBM_GUI_ACTIVE (separate passes) -- python scripts/serial_kernels.py 22 2|3",
       "reading": "fractions are per wave (of SQ_WAVE_CYCLES); clock_MHz = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration: the clock the chip's "
                  "power management holds during that kernel",
       "kernels": {}}
for k, c in sq.items():
    wc = c.get("SQ_WAVE_CYCLES", [0])[0]
    if wc <= 0 or not k.startswith("k_"):
        continue
    g = lambda n: c.get(n, [0.0])[0]
    res["kernels"][k] = {"launches": c["SQ_WAVE_CYCLES"][1], "active_inst_valu_frac": g("SQ_ACTIVE_INST_VALU") / wc, "active_inst_any_frac": g("SQ_ACTIVE_INST_ANY") / wc,
                         "wait_inst_any_frac": g("SQ_WAIT_INST_ANY") / wc, "wait_any_frac": g("SQ_WAIT_ANY") / wc, "valu_insts_per_wave": g("SQ_INSTS_VALU") / max(1.0, g("SQ_WAVES"))}
for k, c in clk.items():
    if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"][2] > 0 and k.startswith("k_"):
        res["kernels"].setdefault(k, {})["clock_MHz"] = c["GRBM_GUI_ACTIVE"][0] / c["GRBM_GUI_ACTIVE"][2] / 8 * 1e3
        res["kernels"][k]["avg_launch_us"] = c["GRBM_GUI_ACTIVE"][2] / c["GRBM_GUI_ACTIVE"][1] / 1e3
keep = {k: v for k, v in res["kernels"].items() if v.get("avg_launch_us", 0) > 50 or "valu_insts_per_wave" in v}
res["kernels"] = keep
json.dump(res, open(os.path.join(root, "r06_pmc_sq_clock.json"), "w"), indent=1)
for k, v in sorted(keep.items()):
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
