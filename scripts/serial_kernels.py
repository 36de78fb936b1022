"""Every kernel of the 2^22 step ALONE on the GPU (nothing overlaps: one share component per MSM call, one context), several launches
each — the workload of the serial rocprofv3 traces in profiles/ (kernel stats, PMC traffic, SQ counters, per-kernel clock).
usage: python scripts/serial_kernels.py [log_m=22] [reps=5] [bn254|bls12_381]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0); ctx = cg.Context(0)
stream = torch.cuda.Stream(device=dev); ctx.set_stream(stream.cuda_stream); torch.cuda.set_stream(stream)
C = cg.BLS12_381 if len(sys.argv) > 3 and sys.argv[3] == "bls12_381" else cg.BN254
w = bench.Workload(ctx, log_m, dev, 0, 1, precompute=-1, curve=C)
for _ in range(reps):
    ctx.spmv_csr(C, w.rpA, w.colA, w.coA, w.nc, w.pub, w.n_inputs, 0, w.wa, w.wb, w.aa, w.ab); ctx.sync()
    ctx.spmv_csr(C, w.rpB, w.colB, w.coB, w.nc, w.pub, w.n_inputs, 0, w.wa, w.wb, w.ba, w.bb); ctx.sync()
    ctx.vec_rep3_mul_local(C, w.ca, w.aa, w.ab, w.ba, w.bb, w.mask1, w.m); ctx.sync()
    ctx.ntt_dev(C, [w.aa], w.m, w.omega, inverse=True, coset_gen=w.coset_g); ctx.sync()
    ctx.ntt_dev(C, [w.aa], w.m, w.omega); ctx.sync()
    ctx.vec_sub(C, w.ha, w.ca, w.aa, w.m); ctx.sync()
    for key in (("h", 0, 1), ("b2", 0, 1)):
        bases, lo, hi = w.tables[key]
        sc = [(w.ha if key[0] == "h" else w.wa)[lo:hi]]
        ctx.msm_end(ctx.msm_dev_begin_multi([bases], sc, hi - lo)[0]); ctx.sync()
print("serial kernels done")
