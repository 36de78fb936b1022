"""A/B of the REP3 mul_vec exchange in the host mirror: one synchronous message per vector vs 4 MiB chunks through
page-locked rings on the copy streams, overlapped with the NTTs that do not depend on the product.  Three parties share the GPU.
usage: python scripts/rep3_exchange_ab.py [log_m]"""
import importlib, os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
cg = importlib.import_module("collaborative-circom_amd")
import oracle_lib as orc
from oracle_lib import BN254, FR
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20; m = 1 << log_m
d = tempfile.mkdtemp(); zp, wp = os.path.join(d, "s.zkey"), os.path.join(d, "s.wtns")
orc.make_synthetic(BN254, log_m, 5, zp, wp, threads=min(64, os.cpu_count() or 8))
w = orc.read_wtns(BN254, wp); rng = np.random.default_rng(1)
a = orc.random_field(BN254, FR, w.shape[0] - 2, rng); b = orc.random_field(BN254, FR, w.shape[0] - 2, rng)
c = orc.field_op(BN254, FR, "sub", orc.field_op(BN254, FR, "sub", w[2:], a), b)
wa, wb = [a, b, c], [c, a, b]
streams = [orc.random_field(BN254, FR, 2 * m + 4, rng) for _ in range(3)]
for rnd in range(2):
    for name, thr in (("one synchronous message", 1 << 40), ("chunked, asynchronous", 1 << 19)):
        cg.host_set_option(cg.HOST_OPT_XCHG_ASYNC_MIN, thr)
        ts = []
        for _ in range(3):
            t0 = time.time(); cg.prove_rep3(BN254, zp, w[:2], wa, wb, streams); ts.append(round((time.time() - t0) * 1e3, 1))
        print(f"2^{log_m} {name}: file -> 3 proofs {ts} ms", flush=True)
