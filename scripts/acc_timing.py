"""The bucket accumulation alone (library events on the kernel's own stream): G1 and G2 launches at several sizes, ms per launch and the achieved
multiply-add rate.  A/B of the mixed addition's product form: COGROTH16_HIP_LIB selects another build of the library.
usage: python scripts/acc_timing.py [sizes=19,20,22] [groups=g1,g2] [curve=bn254]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "19,20,22").split(",")]
groups = (sys.argv[2] if len(sys.argv) > 2 else "g1,g2").split(",")
curve = cg.BLS12_381 if len(sys.argv) > 3 and sys.argv[3].startswith("bls") else cg.BN254
dev = torch.device("cuda", 0); torch.cuda.set_device(0); ctx = cg.Context(0)
g = torch.Generator(device=dev); g.manual_seed(7)
print(f"# library: {cg.LIB_PATH}")
for lg in sizes:
    n = 1 << lg
    sc = bench.rand_fr(n, dev, g, curve)
    for grp in groups:
        group = cg.G1 if grp == "g1" else cg.G2
        bases = ctx.synth_bases(curve, group, 1, n); ctx.precompute_bases(bases, 0)
        ctx.msm_end(ctx.msm_dev_begin_multi([bases], [sc], n)[0]); ctx.sync()
        ctx.stats_enable(True); ctx.stats(reset=True)
        reps = 5
        for _ in range(reps): ctx.msm_end(ctx.msm_dev_begin_multi([bases], [sc], n)[0])
        ctx.sync(); st = ctx.stats(reset=True); ctx.stats_enable(False)
        key = "msm_acc_g1_ms" if grp == "g1" else "msm_acc_g2_ms"; calls = st[key.replace("_ms", "_calls")]
        ms = st[key] / max(1, calls)
        print(f"2^{lg} {grp}: accumulate {ms:.4f} ms per launch ({calls} launches), sort {st['msm_sort_ms'] / reps:.3f} ms, reduce {st['msm_reduce_ms'] / reps:.3f} ms", flush=True)
        bases.release()
