"""MSM time for witness-like scalar vectors (the plain driver on a real circuit: many 0 and 1 values, small integers) against uniform
full-width scalars (what REP3 shares are).  usage: python scripts/nonuniform_timing.py [log_n=22]"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
dev = torch.device("cuda", 0); ctx = cg.Context(0)
g = torch.Generator(device=dev); g.manual_seed(3)
uni = bench.rand_fr(n, dev, g)
one = torch.from_numpy(np.array(ctx.fr_from_canonical(cg.BN254, np.array([[1, 0, 0, 0]], dtype=np.uint64))).view(np.int64)).to(dev) if hasattr(ctx, "fr_from_canonical") else None
def mont_small(vals):
    raw = np.zeros((len(vals), 4), dtype=np.uint64); raw[:, 0] = vals
    out = np.zeros_like(raw); cg._chk(cg.load().cg_fr_from_canonical(cg.BN254, cg._hp(raw), cg._hp(out), len(vals))); return out
sel = torch.rand(n, device=dev, generator=g)
small = torch.from_numpy(mont_small(np.arange(256, dtype=np.uint64)).view(np.int64)).to(dev)
mix = uni.clone()
mix[sel < 0.5] = 0                                   # 50 % zeros
m1 = (sel >= 0.5) & (sel < 0.8); mix[m1] = small[1]  # 30 % ones
m2 = (sel >= 0.8) & (sel < 0.9); mix[m2] = small[torch.randint(2, 256, (int(m2.sum()),), device=dev, generator=g)]   # 10 % bytes; 10 % stay full width
for group in (0, 1):
    bases = ctx.synth_bases(cg.BN254, group, 1, n)
    for pre in (False, True):
        if pre: ctx.precompute_bases(bases, 0)
        for name, sc in (("uniform", uni), ("witness-like (50% 0, 30% 1, 10% bytes, 10% full)", mix)):
            run = lambda: ctx.msm_end(ctx.msm_dev_begin_multi([bases], [sc], n)[0])
            r = run(); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(3): run()
            ctx.sync()
            print(f"G{group + 1} 2^{lg} {'precomputed' if pre else 'classic'} {name}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
    bases.release()
