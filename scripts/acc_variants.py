"""GPU A/B of the bucket-accumulation kernel variants (CG_ACC_VARIANT, read per call by msm_accumulate_reduce):
per-launch time of k_msm_accumulate (HIP events of the library, `msm_acc_g1_ms` / `msm_acc_g2_ms`) on one 2^LOG-point table with
precomputed window tables, alone on the GPU.  Every variant must return the same two points as variant 0.
usage: python scripts/acc_variants.py [log_n=22] [variants=0,1,2,3] [groups=0,1] [reps=4]"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
variants = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3").split(",")]
groups = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda", 0)
ctx = cg.Context(0)
stream = torch.cuda.Stream(device=dev); ctx.set_stream(stream.cuda_stream); torch.cuda.set_stream(stream)
g = torch.Generator(device=dev); g.manual_seed(7)
n = 1 << lg
sc = [bench.rand_fr(n, dev, g)]      # one component per call: digits/sort, accumulate, reduce run one after the other (nothing overlaps)
for group in groups:
    bases = ctx.synth_bases(cg.BN254, group, 1, n)
    ctx.precompute_bases(bases, 0)
    ref = None
    for v in variants:
        os.environ["CG_ACC_VARIANT"] = str(v)
        def run():
            tk = ctx.msm_dev_begin_multi([bases], sc, n)
            return [ctx.msm_end(t) for t in tk]
        out = run(); ctx.sync()
        aff = [cg.point_to_affine(cg.BN254, group, out[0][j]) for j in range(len(sc))]
        if ref is None:
            ref = aff
        same = all(np.array_equal(a, b) for a, b in zip(aff, ref))
        ctx.stats_enable(True); ctx.stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        ctx.sync()
        wall = (time.perf_counter() - t0) / reps * 1e3
        st = ctx.stats(reset=True); ctx.stats_enable(False)
        key = "msm_acc_g1" if group == 0 else "msm_acc_g2"
        print(f"G{group + 1} 2^{lg} variant {v}: acc {st[key + '_ms'] / max(1, st[key + '_calls']):.3f} ms/launch ({int(st[key + '_calls'])} launches), "
              f"sort {st['msm_sort_ms'] / reps:.2f} ms, reduce {st['msm_reduce_ms'] / reps:.2f} ms, wall {wall:.2f} ms per MSM, same_as_v0={same}", flush=True)
    bases.release()
os.environ.pop("CG_ACC_VARIANT", None)
