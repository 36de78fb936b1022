"""GPU sweep: classic (no precomputed tables) MSM time per window size c and table size, 2 share components."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
dev = torch.device("cuda", 0)
ctx = cg.Context(0)
stream = torch.cuda.Stream(device=dev); ctx.set_stream(stream.cuda_stream); torch.cuda.set_stream(stream)
g = torch.Generator(device=dev); g.manual_seed(1)
for group in (cg.G1, cg.G2):
    for lg in (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12,16,20").split(",")):
        n = 1 << lg
        sc = [bench.rand_fr(n, dev, g), bench.rand_fr(n, dev, g)]
        bases = ctx.synth_bases(cg.BN254, group, 1, n)
        row = []
        for c in [0] + list(range(6, 18)):
            ctx.set_msm_window(c)
            def run():
                tk = ctx.msm_dev_begin_multi([bases], sc, n)
                return [ctx.msm_end(t) for t in tk]
            run(); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(5): run()
            ctx.sync()
            row.append((c, round((time.perf_counter() - t0) / 5 * 1e3, 2)))
        print("G%d 2^%d" % (group + 1, lg), row, flush=True)
        bases.release()
