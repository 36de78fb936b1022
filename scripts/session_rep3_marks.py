import importlib, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, "/root/repo")
cg = importlib.import_module("collaborative-circom_amd")
import bench
log_m = 22
dev = torch.device("cuda", 0); ctx = cg.Context(0)
d = tempfile.mkdtemp(); zp, wp = os.path.join(d, "s.zkey"), os.path.join(d, "s.wtns")
cg.host_synth_circuit(cg.BN254, log_m, 5, zp, wp)
w = cg.host_read_wtns(cg.BN254, wp); m = 1 << log_m; n_aux = m - 2
g = torch.Generator(device=dev); g.manual_seed(1)
host = lambda t: t.cpu().numpy().view(np.uint64)
da, db = bench.rand_fr(n_aux, dev, g), bench.rand_fr(n_aux, dev, g)
dw = torch.from_numpy(np.ascontiguousarray(w[2:]).view(np.int64)).to(dev); dc = torch.empty_like(dw)
ctx.vec_sub(cg.BN254, dc, dw, da, n_aux); ctx.vec_sub(cg.BN254, dc, dc, db, n_aux); ctx.sync()
pin = lambda x: (lambda p: (p.__setitem__(slice(None), x), p)[1])(ctx.host_alloc(x.shape))
a, b, c = pin(host(da)), pin(host(db)), pin(host(dc))
streams = [pin(host(bench.rand_fr(2 * m + 4, dev, g))) for _ in range(3)]
ses = cg.ProvingSession(cg.BN254, zp, precompute=True, validate=False)
ses.prove_rep3(w[:2], [a, b, c], [c, a, b], streams, solo=False)
os.environ["CGH_TIMING"] = "1"
print("---- timed", file=sys.stderr)
out = ses.prove_rep3(w[:2], [a, b, c], [c, a, b], streams)
print("three", out[1] * 1e3, "solo", out[2] * 1e3)
