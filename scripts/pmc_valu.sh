#!/bin/bash
# SQ issue/stall counters of the MSM kernels (one serial step, so launches do not share the CUs): gpurun_out/pmc_valu/summary.json
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_valu; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d $O/raw -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --one-context > $O/run.log 2>&1
cd $R
python - <<PY
import csv, glob, json, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob("$O/raw/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        m = re.search(r"(k_\w+)", k)
        if not m: continue
        name = m.group(1) + ("<G2>" if "Fp2" in k else "<G1>" if "FqP" in k else "")
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"]); n[name] += 1
out = {}
for k, c in acc.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0: continue
    launches = n[k] / max(1, len(c))
    out[k] = {"launches": launches, "wave_cycles_per_launch": wc / launches,
              "active_inst_any_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / wc, "active_inst_valu_frac": c.get("SQ_ACTIVE_INST_VALU", 0) / wc,
              "wait_inst_any_frac": c.get("SQ_WAIT_INST_ANY", 0) / wc, "wait_any_frac": c.get("SQ_WAIT_ANY", 0) / wc,
              "valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0) / max(1.0, c.get("SQ_WAVES", 1.0))}
keep = {k: v for k, v in out.items() if k.startswith(("k_msm_accumulate", "k_ntt", "k_msm_reduce", "k_items", "k_part_scatter"))}
json.dump(keep, open("$O/summary.json", "w"), indent=1)
print(json.dumps(keep, indent=1)[:2500])
PY
find $O/raw -name '*.csv' -size +1M -delete
