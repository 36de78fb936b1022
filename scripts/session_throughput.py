"""Proofs per second of the host mirror on an open proving session (zkey resident, per-window precomputed tables): the plain driver,
three co-located REP3 parties, and ONE REP3 party alone on the GPU (its received messages replayed) — the drop-in's counterpart of
bench.py's step.  usage: python scripts/session_throughput.py [log_m ...]"""
import importlib, os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
cg = importlib.import_module("collaborative-circom_amd")
import oracle_lib as orc
from oracle_lib import BN254, FR

hctx = cg.Context(0)
for log_m in [int(x) for x in sys.argv[1:]] or [20]:
    m = 1 << log_m
    d = tempfile.mkdtemp(); zp, wp = os.path.join(d, "s.zkey"), os.path.join(d, "s.wtns")
    orc.make_synthetic(BN254, log_m, 5, zp, wp, threads=min(64, os.cpu_count() or 8))
    w = orc.read_wtns(BN254, wp); rng = np.random.default_rng(1)
    z = orc.ZKey(BN254, zp)
    v1 = z.points("vk_g1"); v2 = z.points("vk_g2")
    vk = {"alpha1": v1[0], "beta2": v2[0], "gamma2": v2[1], "delta2": v2[2], "ic": z.points("ic")}
    for pre in (False, True):
        t0 = time.time(); ses = cg.ProvingSession(BN254, zp, precompute=pre); t_open = time.time() - t0
        r, s = orc.random_field(BN254, FR, 2, rng)
        ses.prove_plain(w, r, s)
        proof, t_plain = min((ses.prove_plain(w, r, s) for _ in range(3)), key=lambda x: x[1])
        a = orc.random_field(BN254, FR, w.shape[0] - 2, rng); b = orc.random_field(BN254, FR, w.shape[0] - 2, rng)
        c = orc.field_op(BN254, FR, "sub", orc.field_op(BN254, FR, "sub", w[2:], a), b)
        streams = [orc.random_field(BN254, FR, 2 * m + 4, rng) for _ in range(3)]
        ses.prove_rep3(w[:2], [a, b, c], [c, a, b], streams, solo=False)
        proofs, t3, t1 = ses.prove_rep3(w[:2], [a, b, c], [c, a, b], streams)
        # the same with the share vectors and randomness streams in page-locked memory (what a caller that allocates them with
        # cg_host_alloc gets): the library then copies straight from them
        pin = lambda x: (lambda p: (p.__setitem__(slice(None), x), p)[1])(hctx.host_alloc(x.shape))
        pa, pb, pc = pin(a), pin(b), pin(c); pstreams = [pin(x) for x in streams]
        proofs_p, t3p, t1p = ses.prove_rep3(w[:2], [pa, pb, pc], [pc, pa, pb], pstreams)
        assert (proofs_p == proofs).all()
        for x in [pa, pb, pc] + pstreams: hctx.host_free(x)
        ok = orc.verify(BN254, vk, w[1:2], proof) and orc.verify(BN254, vk, w[1:2], proofs[0]) and (proofs[0] == proofs[1]).all() and (proofs[1] == proofs[2]).all()
        ses.close()
        print(f"2^{log_m} session ({'precomputed window tables' if pre else 'plain tables'}; open {t_open * 1e3:.0f} ms): plain prove {t_plain * 1e3:.1f} ms "
              f"({z.num_constraints / t_plain / 1e6:.1f} M constraints/s); REP3: three parties sharing the GPU {t3 * 1e3:.1f} ms, one party alone {t1 * 1e3:.1f} ms "
              f"({z.num_constraints / t1 / 1e6:.1f} M constraints/s), with page-locked shares and streams {t1p * 1e3:.1f} ms ({z.num_constraints / t1p / 1e6:.1f} M constraints/s); verify {'ok' if ok else 'FAILED'}", flush=True)
