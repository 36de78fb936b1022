import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
cg = importlib.import_module("collaborative-circom_amd")
import bench
dev = torch.device("cuda", 0); ctx = cg.Context(0)
g = torch.Generator(device=dev); g.manual_seed(1)
for lg in (20, 22, 24):
    m = 1 << lg
    v = [bench.rand_fr(m, dev, g) for _ in range(2)]
    r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    zt = pow(5, (r - 1) >> 28, r); root = lambda k: pow(zt, 1 << (28 - k), r)
    mont = lambda x: np.array([((x << 256) % r >> (64 * j)) & (2**64 - 1) for j in range(4)], dtype=np.uint64)
    om, cs = mont(root(lg)), mont(root(lg + 1))
    for name, kw in (("forward", {}), ("inverse+coset", dict(inverse=True, coset_gen=cs))):
        ctx.ntt_dev(cg.BN254, v, m, om, **kw); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(10): ctx.ntt_dev(cg.BN254, v, m, om, **kw)
        ctx.sync()
        print(f"2^{lg} {name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per transform (2 vectors per call)", flush=True)
