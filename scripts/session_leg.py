"""bench.py's entry leg alone (one REP3 party with its draws inside the call, three parties, plain driver, Shamir twin): for A/B runs of
host-side scheduling knobs.  usage: python scripts/session_leg.py [log_m=22 | poseidon] [proofs=5]"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
fixture = len(sys.argv) > 1 and sys.argv[1] == "poseidon"      # the reference's own bench circuit (tests/golden/groth16/bn254/poseidon, m = 256)
log_m = 0 if fixture else int(sys.argv[1]) if len(sys.argv) > 1 else 22
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0); torch.cuda.set_device(0); ctx = cg.Context(0)
fx = os.path.join(ROOT, "tests", "golden", "groth16", "bn254", "poseidon")
curve = cg.BLS12_381 if os.environ.get("CURVE", "").startswith("bls") else cg.BN254             # CURVE=bls12_381: the second curve
if fixture and curve == cg.BLS12_381: fx = fx.replace("bn254", "bls12_381")
out = bench.entry_leg(ctx, log_m, dev, proofs, 1, curve=curve, extras=bool(os.environ.get("EXTRAS")) or (not os.environ.get("NO_EXTRAS") and not fixture),
                      files=(os.path.join(fx, "circuit.zkey"), os.path.join(fx, "witness.wtns")) if fixture else None)
res = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items() if k.endswith("_ms") or k.startswith("ms_per_proof")}
sh = out.get("shamir_party") or {}
res.update({"shamir_" + k: round(x, 2) for k, x in sh.items() if k.endswith("_ms")})
res["each"] = out.get("ms_inner_each")
print(json.dumps(res))
