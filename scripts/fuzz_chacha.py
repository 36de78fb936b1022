"""Randomised differential test of the device draws (cg_chacha12_fr_rand_dev) against the oracle's restatement and the host library's:
seeds, word positions (aligned, unaligned, around 2^32 and 2^36 block counters), sizes, both curves; chained calls continue each other.
usage: python scripts/fuzz_chacha.py [seconds] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
cg = importlib.import_module("collaborative-circom_amd")
import oracle_lib as orc
from oracle_lib import BN254, BLS12_381

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = cg.Context(0)
t0 = time.time(); cases = 0; draws = 0
while time.time() - t0 < budget:
    curve = BN254 if rng.random() < 0.6 else BLS12_381
    seed = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
    base = int(rng.choice([0, 0, 16, 1 << 20, (1 << 32) - 64, (1 << 36) - 64, (1 << 40), (1 << 62)]))
    pos = base + int(rng.integers(0, 128)) * int(rng.choice([1, 8]))
    n = int(rng.choice([1, 2, 3, 63, 64, 255, 256, 257, 511, 513, 1000, 4096, 4097, 65537, 300000]))
    want, wa = orc.chacha12_fr_rand(curve, seed, pos, n)
    # in one call, and as two chained calls (the second starts where the first reports)
    buf, ga = ctx.chacha12_fr_rand(curve, seed, pos, n); got = buf.download((n, 4)); buf.free()
    assert ga == wa and (got == want).all(), ("single", curve, pos, n)
    k = int(rng.integers(0, n + 1))
    b1, mid = ctx.chacha12_fr_rand(curve, seed, pos, k); g1 = b1.download((k, 4)) if k else np.zeros((0, 4), dtype=np.uint64); b1.free()
    b2, end = ctx.chacha12_fr_rand(curve, seed, mid, n - k); g2 = b2.download((n - k, 4)) if n - k else np.zeros((0, 4), dtype=np.uint64); b2.free()
    assert end == wa and (np.concatenate([g1, g2]) == want).all(), ("chained", curve, pos, n, k)
    if n <= 4097:
        hw, ha = cg.chacha12_fr_rand_host(curve, seed, pos, n)
        assert ha == wa and (hw == want).all(), ("host", curve, pos, n)
    cases += 1; draws += 2 * n
print(f"{cases} cases, {draws} draws compared in {time.time() - t0:.0f} s: device == oracle == host library (values and word positions)")
