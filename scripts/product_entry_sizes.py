"""The product's session entry (cgh_session_prove_rep3_party_ex: host buffers in, proof out, the party's ChaCha12 draws and PCIe inside the
timed call) at several circuit sizes, with bench.py's own entry leg.  usage: python scripts/product_entry_sizes.py [--curve bls12_381] [log_m ...]"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
argv = sys.argv[1:]
curve = cg.BN254
if "--curve" in argv:
    i = argv.index("--curve"); curve = cg.BLS12_381 if argv[i + 1] == "bls12_381" else cg.BN254; del argv[i:i + 2]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = cg.Context(0)
for log_m in [int(x) for x in argv] or [20, 22]:
    s = bench.entry_leg(ctx, log_m, dev, 5, 1, curve, extras=True)
    nc = (1 << log_m) - 2
    print(f"{s['curve']} 2^{log_m}: one REP3 party {s['ms_per_proof']:.1f} ms per proof over {s['proofs']} proofs (best {s['ms_per_proof_min_inner']:.1f}; {s['value'] / 1e6:.1f} M constraints/s), "
          f"draws on one host thread instead: {s.get('ms_per_proof_host_draws')} ms; plain driver {s.get('plain_driver_ms')} ms; three parties on one GPU {s['rep3_three_parties_one_gpu_ms']:.1f} ms; "
          f"zkey {s['zkey']['file_bytes'] / 1e9:.2f} GB generated in {s['zkey']['generate_s']:.1f} s, session open {s['zkey']['session_open_s']:.1f} s; proofs agree: {s['three_parties_agree']}", flush=True)
    sh = s.get("shamir_party") or {}
    if "party_ms" in sh:
        print(f"    one Shamir party (t = 1 of 3, seeded entry, reference protocol): {sh['party_ms']:.1f} ms, king {sh['king_ms']:.1f} ms; three co-located {sh['three_parties_one_gpu_ms']:.1f} ms", flush=True)
    elif sh: print("    shamir leg failed:", sh, flush=True)
