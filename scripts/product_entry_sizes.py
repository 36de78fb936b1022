"""The product's session entry points (cgh_session_prove_plain / cgh_session_prove_rep3_party: host buffers in, proof out, PCIe inside the
timed call) at several circuit sizes, with bench.py's own session leg.  usage: python scripts/product_entry_sizes.py [log_m ...]"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = cg.Context(0)
for log_m in [int(x) for x in sys.argv[1:]] or [20, 22]:
    s = bench.session_leg(ctx, log_m, dev)
    nc = (1 << log_m) - 2
    print(f"2^{log_m}: plain {s['plain_ms']:.1f} ms ({nc / s['plain_ms'] / 1e3:.1f} M constraints/s); one REP3 party {s['rep3_party_ms']:.1f} ms mean / {s['rep3_party_ms_min']:.1f} min "
          f"({nc / s['rep3_party_ms'] / 1e3:.1f} M constraints/s); three parties on one GPU {s['rep3_three_parties_one_gpu_ms']:.1f} ms; zkey {s['zkey']['file_bytes'] / 1e9:.2f} GB generated in "
          f"{s['zkey']['generate_s']:.1f} s, session open {s['zkey']['session_open_s']:.1f} s; proofs agree: {s['three_parties_agree']}", flush=True)
    c = s.get("chacha12_randomness") or {}
    if "rep3_party_ms_device_draws" in c:
        print(f"    with the party's ChaCha12 draws inside the call: {c['rep3_party_ms_device_draws']:.1f} ms drawn on the GPU, {c['rep3_party_ms_host_draws']:.1f} ms drawn on one host thread "
              f"({c['draws_per_proof']} draws per proof)", flush=True)
    elif c: print("    chacha leg failed:", c, flush=True)
    sh = s.get("shamir_party") or {}
    if "party_ms" in sh:
        print(f"    one Shamir party (t = 1 of 3, seeded entry, reference protocol): {sh['party_ms']:.1f} ms, king {sh['king_ms']:.1f} ms; three co-located {sh['three_parties_one_gpu_ms']:.1f} ms", flush=True)
    elif sh: print("    shamir leg failed:", sh, flush=True)
    v = s.get("additive_h_variant") or {}
    if "rep3_party_ms" in v:
        print(f"    opt-in additive-quotient variant: one REP3 party {v['rep3_party_ms']:.1f} ms mean / {v['rep3_party_ms_min']:.1f} min; three parties on one GPU "
              f"{v['rep3_three_parties_one_gpu_ms']:.1f} ms; same proofs as the reference protocol: {v['same_proofs_as_reference_protocol']}", flush=True)
    elif v: print("    variant failed:", v, flush=True)
