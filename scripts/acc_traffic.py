"""One G1 bucket accumulation at the bench line's size, alone on the GPU: the workload bench.py puts under `rocprofv3 --kernel-trace --pmc
FETCH_SIZE` / `--pmc WRITE_SIZE` (two separate passes, MI355X_MICROARCH.md's HBM recipe) to fill `roofline.traffic` in the same run.
usage: python scripts/acc_traffic.py [log_m=22] [reps=2]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0); torch.cuda.set_device(0); ctx = cg.Context(0)
n = 1 << log_m
g = torch.Generator(device=dev); g.manual_seed(7)
sc = bench.rand_fr(n, dev, g)
bases = ctx.synth_bases(cg.BN254, cg.G1, 1, n)
ctx.precompute_bases(bases, 0)
for _ in range(reps):
    ctx.msm_end(ctx.msm_dev_begin_multi([bases], [sc], n)[0]); ctx.sync()
print("accumulation done")
