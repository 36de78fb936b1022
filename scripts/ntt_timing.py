"""Transforms alone on the GPU (library events on the kernels' own stream): one forward transform and one iNTT -> coset shift -> NTT pair per
size, with the achieved multiply-add rate (162 v_mad per butterfly, m/2 log2 m butterflies per transform) against the 30 Tmad/s a pure
multiply-add loop sustains.  A/B of the butterflies' product form: COGROTH16_HIP_LIB selects another build of the library.
usage: python scripts/ntt_timing.py [sizes=16,20,22,24] [curve=bn254]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,20,22,24").split(",")]
curve = cg.BLS12_381 if len(sys.argv) > 2 and sys.argv[2].startswith("bls") else cg.BN254
dev = torch.device("cuda", 0); ctx = cg.Context(0)
g = torch.Generator(device=dev); g.manual_seed(1)
print(f"# library: {cg.LIB_PATH}")
for lg in sizes:
    m = 1 << lg
    r, two_adicity, _ = bench.FR[curve]                        # snarkjs roots (co-circom-snarks/src/lib.rs:208-221), as bench.Workload computes them
    zt = pow(5, (r - 1) >> two_adicity, r)
    root = lambda k: pow(zt, 1 << (two_adicity - k), r)
    mont = lambda x: np.array([((x << 256) % r >> (64 * j)) & (2**64 - 1) for j in range(4)], dtype=np.uint64)
    omega, coset_g = mont(root(lg)), mont(root(lg + 1))
    v = bench.rand_fr(m, dev, g, curve)
    def timed(fn, reps=10):
        fn(); ctx.sync(); ctx.stats_enable(True); ctx.stats(reset=True)
        for _ in range(reps): fn()
        ctx.sync(); st = ctx.stats(reset=True); ctx.stats_enable(False)
        return st["ntt_ms"] / reps
    t1 = timed(lambda: ctx.ntt_dev(curve, [v], m, omega))
    t2 = timed(lambda: ctx.ntt_coset_pair_dev(curve, [v], m, omega, coset_g))
    mads = 162.0 * (m / 2) * lg
    print(f"2^{lg}: transform {t1:.4f} ms = {mads / (t1 * 1e-3) / 1e12:.2f} Tmad/s = {mads / (t1 * 1e-3) / 1e12 / 30.0:.3f} of 30; "
          f"pair {t2:.4f} ms = {2 * mads / (t2 * 1e-3) / 1e12 / 30.0:.3f}; {64.0 * m / (t1 * 1e-3) / 1e9:.0f} GB/s algorithmic", flush=True)
