"""End-to-end proof latency through the host mirror (C++ drivers over the C ABI): zkey file -> proof, plain driver and three REP3
parties sharing one GPU, on a synthetic satisfiable circuit.  usage: [E2E_CURVE=bls12_381] [E2E_SHAMIR=0] python scripts/e2e_proof_latency.py [log_m ...]"""
import importlib, os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
cg = importlib.import_module("collaborative-circom_amd")
import oracle_lib as orc
from oracle_lib import BN254, BLS12_381, FR
if os.environ.get("E2E_CURVE", "bn254") == "bls12_381": BN254 = BLS12_381     # same script on the second curve

for log_m in [int(x) for x in sys.argv[1:]] or [16, 18]:
    d = tempfile.mkdtemp()
    zp, wp = os.path.join(d, "s.zkey"), os.path.join(d, "s.wtns")
    t0 = time.time(); orc.make_synthetic(BN254, log_m, 5, zp, wp, threads=min(64, os.cpu_count() or 8)); t_gen = time.time() - t0
    w = orc.read_wtns(BN254, wp)
    rng = np.random.default_rng(1)
    r, s = orc.random_field(BN254, FR, 2, rng)
    cg.prove_plain(BN254, zp, w, r, s)                                   # warm-up (module load, first-touch)
    t0 = time.time(); proof = cg.prove_plain(BN254, zp, w, r, s); t_plain = time.time() - t0
    a = orc.random_field(BN254, FR, w.shape[0] - 2, rng); b = orc.random_field(BN254, FR, w.shape[0] - 2, rng)
    c = orc.field_op(BN254, FR, "sub", orc.field_op(BN254, FR, "sub", w[2:], a), b)
    wa, wb = [a, b, c], [c, a, b]
    m = 1 << log_m
    streams = [orc.random_field(BN254, FR, 2 * m + 4, rng) for _ in range(3)]
    t0 = time.time(); proofs = cg.prove_rep3(BN254, zp, w[:2], wa, wb, streams); t_rep3 = time.time() - t0
    z = orc.ZKey(BN254, zp)
    v1 = z.points("vk_g1"); v2 = z.points("vk_g2")
    vk = {"alpha1": v1[0], "beta2": v2[0], "gamma2": v2[1], "delta2": v2[2], "ic": z.points("ic")}
    ok = orc.verify(BN254, vk, w[1:2], proof) and orc.verify(BN254, vk, w[1:2], proofs[0])
    t_sh = t_shp = float("nan")
    if os.environ.get("E2E_SHAMIR", "1") != "0":
        n, t = 3, 1
        wits = orc.shamir_share(BN254, w[2:], n, t, rng)
        need = (2 * m + 4) // (1024 * (t + 1)) + 1
        amount = (2 * m + 8) // (t + 1) + 1
        sstreams = [orc.random_field(BN254, FR, (need * 1024 + amount) * (1 + 3 * t) + t * (2 * m + 8), rng) for _ in range(n)]
        t0 = time.time(); sproofs = cg.prove_shamir(BN254, zp, n, t, w[:2], wits, sstreams); t_sh = time.time() - t0
        ok = ok and orc.verify(BN254, vk, w[1:2], sproofs[0])
        t0 = time.time(); sproofs = cg.prove_shamir(BN254, zp, n, t, w[:2], wits, sstreams, preprocess=amount); t_shp = time.time() - t0
        ok = ok and orc.verify(BN254, vk, w[1:2], sproofs[0])
    print(f"2^{log_m}: zkey {os.path.getsize(zp) / 1e6:.0f} MB (generated in {t_gen:.1f} s); file -> proof: plain {t_plain * 1e3:.0f} ms, "
          f"3 REP3 parties on one GPU {t_rep3 * 1e3:.0f} ms, "
          f"3 Shamir parties (t = 1) {t_sh * 1e3:.0f} ms lazy double sharings / {t_shp * 1e3:.0f} ms preprocessed on the GPU; verify {'ok' if ok else 'FAILED'}", flush=True)
