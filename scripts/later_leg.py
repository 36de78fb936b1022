"""Does a small proof get slower as a LATER leg of a process that has had other sessions?  The Poseidon-fixture party before and after legs at
other sizes, in one process.  usage: python scripts/later_leg.py"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0); ctx = cg.Context(0)
fx = os.path.join(ROOT, "tests", "golden", "groth16", "bn254", "poseidon")
files = (os.path.join(fx, "circuit.zkey"), os.path.join(fx, "witness.wtns"))
def leg(what):
    if isinstance(what, str) and what.startswith("s"):      # "s2": two idle seconds (does the chip's power management explain a slow leg behind a heavy one?)
        import time; time.sleep(float(what[1:])); print(what, "idle", flush=True); return
    if isinstance(what, str) and what.startswith("r"):      # "r16": bench.py's resident leg at that size (two harness contexts), as between the legs of the bench line
        aux = cg.Context(0); r = bench.resident_leg(ctx, aux, dev, int(what[1:]), 10, 2, cg.BN254); aux.sync(); aux.close()
        print(what, "resident step", round(r["ms_per_step"], 2), flush=True); return
    out = bench.entry_leg(ctx, 0 if what == "poseidon" else what, dev, 20 if what == "poseidon" or what <= 16 else 5, 2, extras=False, files=files if what == "poseidon" else None)
    print(what, round(out["ms_per_proof"], 2), "min inner", round(out["ms_per_proof_min_inner"], 2), flush=True)
legs = os.environ.get("LEGS")            # e.g. LEGS=poseidon,16,poseidon for a short traced run
for what in ([x if x == "poseidon" or x[0] in "rs" else int(x) for x in legs.split(",")] if legs else ("poseidon", 16, "poseidon", 20, "poseidon", 22, "poseidon", 16)):
    leg(what)
if os.environ.get("RESIDENT_BETWEEN"):
    r = bench.resident_leg(ctx, cg.Context(0), dev, 20, 5, 2, cg.BN254)
    print("resident 2^20", round(r["ms_per_step"], 2)); leg("poseidon")
if os.environ.get("WARM_TEST"):
    # hypothesis: the slow legs are the GPU's clock state after a lightly loaded stretch.  A 16-legs-then-poseidon pair, once as is and once with
    # 300 ms of dense work right in front of the small leg
    leg(16); leg("poseidon")
    leg(16)
    a = torch.randn(8192, 8192, device=dev); t0 = __import__("time").time()
    while __import__("time").time() - t0 < 0.3: b = a @ a
    torch.cuda.synchronize(); print("(300 ms of dense work)"); leg("poseidon")
