"""Randomised differential test of the MSM paths against the oracle (sizes, groups, windows, precomputed tables, scatter capacities,
skewed / special scalars, repeated / opposite / infinity points, sub-slices).  A quarter of the budget (LARGE_FRAC) goes to LARGE cases,
n = 2^17 .. 2^20 — partition sort, chunk continuation merges, grid reduction, c = 17 .. 20 window tables — on the synthetic table
[(first + i) G], whose exact MSM value is one generator multiplication; three in four of them on BLS12-381.
usage: python scripts/fuzz_msm.py [seconds] [seed] [large_frac]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
cg = importlib.import_module("collaborative-circom_amd")
import oracle_lib as orc
from oracle_lib import BN254, BLS12_381, FR, G1, G2

import bench_check as bc
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
LARGE_FRAC = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
t_large = 0.0; large_cases = {BN254: 0, BLS12_381: 0}


def large_case():
    curve = BLS12_381 if rng.random() < 0.75 else BN254
    group = G1 if rng.random() < 0.6 else G2
    n = int(rng.choice([1 << 17, (1 << 17) + 123, 1 << 18, (1 << 19) + 7, 1 << 20]))
    first = int(rng.integers(1, 1000))
    bases = ctx.synth_bases(curve, group, first, n)
    pre = int(rng.choice([0, -1, 17, 18, 20]))
    if pre: ctx.precompute_bases(bases, max(pre, 0))
    else: ctx.set_msm_window(int(rng.choice([0, 13, 16])))
    k = int(rng.choice([1, 2]))
    scal = []
    for _ in range(k):
        s = orc.random_field(curve, FR, n, rng)
        mode = rng.integers(0, 5)
        if mode == 1: s[rng.random(n) < 0.5] = 0
        if mode == 2: s[rng.random(n) < 0.4] = orc.from_dec(curve, FR, "1")
        if mode == 3: s[rng.random(n) < 0.2] = orc.from_dec(curve, FR, orc.MODULI[(curve, FR)] - 1)
        if mode == 4: s[: n // 2] = s[0]                                              # half the entries of every window in one bucket
        scal.append(s)
    off = int(rng.integers(0, n // 3)) if rng.random() < 0.5 else 0
    m = n - off if rng.random() < 0.5 else int(rng.integers(n // 2, n - off + 1))
    d = [ctx.to_device(s) for s in scal]
    got = ctx.msm_dev(bases, [x.ptr + off * 32 for x in d], m, offset=off)
    for j in range(k):
        want = bc.synth_table_msm(curve, group, scal[j][off:off + m], first + off)
        if not np.array_equal(cg.point_to_affine(curve, group, got[j]), want):
            raise SystemExit(f"MISMATCH (large): curve {curve} group {group} n {n} first {first} off {off} m {m} pre {pre} component {j}")
    ctx.set_msm_window(0); bases.release(); large_cases[curve] += 1

ctx = cg.Context(0)
pool = {}
def points(curve, group):
    if (curve, group) not in pool:
        pool[(curve, group)] = np.stack([orc.generator_mul(curve, group, s) for s in orc.random_field(curve, FR, 96, rng)])
    return pool[(curve, group)]
t0 = time.time(); cases = 0
while time.time() - t0 < budget:
    if t_large < LARGE_FRAC * (time.time() - t0):
        t1 = time.time(); large_case(); t_large += time.time() - t1; cases += 1
        continue
    curve = BN254 if rng.random() < 0.7 else BLS12_381
    group = G1 if rng.random() < 0.6 else G2
    n = int(rng.choice([1, 2, 3, 17, 64, 65, 255, 1000, 2500, 5000, 20000]))
    base = points(curve, group)
    pts = base[rng.integers(0, 96 if rng.random() < 0.7 else 3, size=n)].copy()        # few distinct points: doublings inside buckets
    if rng.random() < 0.5:
        kill = rng.random(n) < rng.choice([0.01, 0.3, 0.9]); pts[kill] = 0            # points at infinity
    k = int(rng.choice([1, 2, 3]))
    scal = []
    for _ in range(k):
        s = orc.random_field(curve, FR, n, rng)
        mode = rng.integers(0, 5)
        if mode == 1: s[:] = s[0]                                                     # one bucket per window
        if mode == 2: s[rng.random(n) < 0.5] = 0
        if mode == 3: s[rng.random(n) < 0.5] = orc.from_dec(curve, FR, "1")
        if mode == 4: s[rng.random(n) < 0.3] = orc.from_dec(curve, FR, orc.MODULI[(curve, FR)] - 1)
        scal.append(s)
    off = int(rng.integers(0, max(1, n // 3))); m = int(rng.integers(1, n - off + 1))
    bases = ctx.register_bases(curve, group, pts)
    pre = int(rng.choice([0, 0, -1, 8, 11, 13, 16, 17, 20]))
    if pre: ctx.precompute_bases(bases, max(pre, 0))
    else: ctx.set_msm_window(int(rng.choice([0, 0, 4, 9, 13, 16])))
    ctx.set_scatter_capacity(int(rng.choice([-1, -1, 0, 3])))
    d = [ctx.to_device(s) for s in scal]
    got = ctx.msm_dev(bases, [x.ptr + off * 32 for x in d], m, offset=off)
    for j in range(k):
        want = orc.msm(curve, group, pts[off:off + m], scal[j][off:off + m], threads=8)
        if not np.array_equal(cg.point_to_affine(curve, group, got[j]), want):
            raise SystemExit(f"MISMATCH: curve {curve} group {group} n {n} off {off} m {m} pre {pre} component {j}")
    ctx.set_msm_window(0); ctx.set_scatter_capacity(-1); bases.release(); cases += 1
print(f"fuzz_msm: {cases} random cases agree with the oracle ({time.time() - t0:.0f} s); of these {large_cases[BLS12_381]} BLS12-381 and "
      f"{large_cases[BN254]} BN254 cases at n = 2^17 .. 2^20 ({t_large:.0f} s)")
