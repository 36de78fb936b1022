"""Measures the planner's cost table of bench.py (ACC_COST): time of one MSM work unit (both share components, accumulate + bucket
reduction, tables precomputed with the automatic window) by table size, pipelined, excluding the shared scalar schedule:
(time of a 4-table call - time of a 1-table call) / 3.  usage: python scripts/unit_cost_table.py"""
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
    row = []
    for lg in (18, 19, 20, 21, 22):
        n = 1 << lg
        sc = [bench.rand_fr(n, dev, g), bench.rand_fr(n, dev, g)]
        tabs = [ctx.synth_bases(cg.BN254, group, 1 + 7 * i, n) for i in range(4)]
        for b in tabs: ctx.precompute_bases(b, 0)
        def run(k, reps=4):
            def once():
                tk = ctx.msm_dev_begin_multi(tabs[:k], sc, n)
                return [ctx.msm_end(t) for t in tk]
            once(); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps): once()
            ctx.sync()
            return (time.perf_counter() - t0) / reps * 1e3
        t1, t4 = run(1), run(4)
        row.append((n / (1 << 20), round((t4 - t1) / 3, 2), round(t1, 2)))
        for b in tabs: b.release()
    print("group", group, "(points in M, ms per extra table, ms for the first table incl. schedule):", row, flush=True)
