"""MSM over a table that is sparse in points (34 % infinity, like the B queries of the poseidon zkey), 2 share components, precomputed
window tables; run with and without CG_NO_COMPACT=1 to see what the registration-time compaction buys."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
dev = torch.device("cuda", 0)
ctx = cg.Context(0)
stream = torch.cuda.Stream(device=dev); ctx.set_stream(stream.cuda_stream); torch.cuda.set_stream(stream)
g = torch.Generator(device=dev); g.manual_seed(1)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
for group in (cg.G1, cg.G2):
    full = ctx.synth_bases(cg.BN254, group, 1, n)
    pts = ctx.bases_download(full, 0, n); full.release()
    rng = np.random.default_rng(5)
    pts[rng.random(n) < 0.34] = 0
    bases = ctx.register_bases(cg.BN254, group, pts)
    ctx.precompute_bases(bases, 0)
    sc = [bench.rand_fr(n, dev, g), bench.rand_fr(n, dev, g)]
    def run():
        tk = ctx.msm_dev_begin_multi([bases], sc, n)
        return [ctx.msm_end(t) for t in tk]
    run(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5): run()
    ctx.sync()
    print("G%d 2^%d, 34%% infinity, compaction %s: %.2f ms" % (group + 1, lg, "off" if os.environ.get("CG_NO_COMPACT") else "on", (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    bases.release()
