"""The scalar-side sort schedule of one 2^log_m MSM alone on the GPU (for rocprofv3 --kernel-trace --stats): digits, MSD partition, counting sort.
usage: python scripts/sort_timing.py [log_m=22] [reps=5]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0); ctx = cg.Context(0)
n = 1 << log_m
g = torch.Generator(device=dev); g.manual_seed(3)
sc = bench.rand_fr(n, dev, g)
bases = ctx.bases_from_scalars(cg.BN254, cg.G1, bench.rand_fr(n, dev, g), n)
ctx.precompute_bases(bases, 0)
ctx.msm_end(ctx.msm_dev_begin_multi([bases], [sc], n)[0]); ctx.sync()
ctx.stats_enable(True); ctx.stats(reset=True)
for _ in range(reps):
    ctx.msm_end(ctx.msm_dev_begin_multi([bases], [sc], n)[0]); ctx.sync()
st = ctx.stats()
print("sort_ms", st["msm_sort_ms"] / reps, "acc_ms", st["msm_acc_g1_ms"] / reps, "reduce_ms", st["msm_reduce_ms"] / reps)
