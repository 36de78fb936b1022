"""Trait-level drop-in (level 1) timed on host buffers: every call uploads its operands from pageable host
memory and downloads its result, exactly what a Rust shim forwarding `mul_vec` / `ifft_in_place` / `fft_in_place` / `msm_public_points`
one call at a time would do.  Constraint evaluation stays on the CPU at this level and is not part of the number.
usage: python scripts/level1_rate.py [log_m]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")

log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 22
m = 1 << log_m
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
rng = np.random.default_rng(1)
def rand_fr(n):
    x = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); x[:, 3] &= np.uint64((1 << 60) - 1); return x    # < 2^252 < r
zt = pow(5, (R - 1) >> 28, R); root = lambda k: pow(zt, 1 << (28 - k), R)
mont = lambda v: np.array([((v << 256) % R >> (64 * j)) & (2**64 - 1) for j in range(4)], dtype=np.uint64)
omega, g = mont(root(log_m)), mont(root(log_m + 1))
ctx = cg.Context(0)
tables = {t: ctx.synth_bases(cg.BN254, grp, first, m) for t, grp, first in (("h", 0, 1), ("l", 0, 3), ("a", 0, 5), ("b1", 0, 7), ("b2", 1, 1))}
for b in tables.values():
    ctx.precompute_bases(b, 0)
a, b = [rand_fr(m), rand_fr(m)], [rand_fr(m), rand_fr(m)]
wit = [rand_fr(m), rand_fr(m)]
mask = rand_fr(m)

def step():
    c = [ctx.vec_rep3_mul_local_host(cg.BN254, a[0], a[1], b[0], b[1], mask), mask]            # mul_vec local part (+ the received component)
    va, vb, vc = (ctx.ntt(cg.BN254, v, omega, inverse=True, coset_gen=g) for v in (a, b, c))  # ifft_in_place + distribute_powers
    va, vb, vc = (ctx.ntt(cg.BN254, v, omega) for v in (va, vb, vc))                           # fft_in_place
    h = [ctx.vec_rep3_mul_local_host(cg.BN254, va[0], va[1], vb[0], vb[1], mask), mask]        # second mul_vec; sub_assign stays on the host
    out = [ctx.msm(tables["h"], h)]
    out += [ctx.msm(tables[t], wit) for t in ("l", "a", "b1", "b2")]
    return out

step()
t0 = time.perf_counter(); K = 3
for _ in range(K): step()
dt = (time.perf_counter() - t0) / K
print(f"level-1 (host buffers, one call at a time) 2^{log_m}: {dt * 1e3:.1f} ms per party step = {(m - 2) / dt / 1e6:.1f} M constraints/s")
