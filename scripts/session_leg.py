"""bench.py's session leg alone (plain proof, one REP3 party, three parties): for A/B runs of host-side scheduling knobs.
usage: python scripts/session_leg.py [log_m=22]"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 22
dev = torch.device("cuda", 0); ctx = cg.Context(0)
out = bench.session_leg(ctx, log_m, dev)
res = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items() if k.endswith("_ms")}
v = out.get("additive_h_variant") or {}
res.update({"variant_" + k: round(x, 2) for k, x in v.items() if k.endswith("_ms") or k.endswith("_ms_min")})
print(json.dumps(res))
