"""Critical path of EVERY device of an N-GPU proving session, timed on one GPU (no multi-GPU box here): the session is opened over N
contexts on GPU 0 and, for d = 0 .. N-1 in turn, CGH_EMULATE_DEVICE=d makes every device but d a no-op — the timed proof is what device d of
N does (its rows of the witness map, its vector pipeline if it owns one, its table slices; d = 0 also the host's folding and the exchange
bookkeeping) while the others would work beside it.  Reported: every device's time and the MAX over devices (VERDICT r5 #4a; round 5
reported the primary only).  What the emulation cannot show: waiting for ANOTHER device's pipeline (a device without one is timed as if the
owners' transforms took no time), and xGMI — peer copies are local HBM copies here.
Needs the planning build of the host library (the release library does not contain the knob):
    make -C collaborative-circom_amd/host KNOBS=1
    COGROTH16_HOST_LIB=collaborative-circom_amd/libcogroth16_host_knobs.so python scripts/multi_device_emulation.py [log_m=22] [worlds=1,2,4,8] [additive]
(additive: sessions opened with CGH_SESSION_ADDITIVE_H, the opt-in protocol variant)"""
import importlib, os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
import bench
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 22
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
additive = len(sys.argv) > 3 and sys.argv[3] == "additive"
assert os.environ.get("COGROTH16_HOST_LIB") or worlds == [1], "set COGROTH16_HOST_LIB to the KNOBS=1 build (the release library ignores CGH_EMULATE_DEVICE: all slices would run on the one GPU)"
dev = torch.device("cuda", 0); ctx = cg.Context(0)
d = tempfile.mkdtemp(); zp, wp = os.path.join(d, "s.zkey"), os.path.join(d, "s.wtns")
cg.host_synth_circuit(cg.BN254, log_m, 5, zp, wp)
w = cg.host_read_wtns(cg.BN254, wp); m = 1 << log_m; n_aux = m - 2
g = torch.Generator(device=dev); g.manual_seed(1)
host = lambda t: t.cpu().numpy().view(np.uint64)
r, s = host(bench.rand_fr(2, dev, g))
da, db = bench.rand_fr(n_aux, dev, g), bench.rand_fr(n_aux, dev, g)
dw = torch.from_numpy(np.ascontiguousarray(w[2:]).view(np.int64)).to(dev); dc = torch.empty_like(dw)
ctx.vec_sub(cg.BN254, dc, dw, da, n_aux); ctx.vec_sub(cg.BN254, dc, dc, db, n_aux); ctx.sync()
pin = lambda x: (lambda p: (p.__setitem__(slice(None), x), p)[1])(ctx.host_alloc(x.shape))
a, b, c = pin(host(da)), pin(host(db)), pin(host(dc))
streams = [pin(host(bench.rand_fr(2 * m + 4, dev, g))) for _ in range(3)]
del da, db, dc, dw
tag = ", additive-quotient variant" if additive else ""
for name, opt in (("WIDE_LOG", cg.HOST_OPT_CTX_WIDE_LOG), ("OFF_MAIN_LOG", cg.HOST_OPT_CTX_OFF_MAIN_LOG)):     # A/B: session contexts' wide / off-main bounds
    if os.environ.get(name): cg.host_set_option(opt, int(os.environ[name])); tag += f", {name}={os.environ[name]}"
if os.environ.get("PRECOMPUTE"): tag += f", window {os.environ['PRECOMPUTE']}"
for world in worlds:
    ses = cg.ProvingSession(cg.BN254, zp, precompute=int(os.environ.get('PRECOMPUTE', 0)) or True, devices=[0] * world, shared_devices=True, validate=False, additive_h=additive)
    plain, party = [], []
    only = os.environ.get("EMU_DEVICES")                      # e.g. EMU_DEVICES=7 under rocprofv3: the trace then ends with device 7's REP3 party
    for d in ([int(x) for x in only.split(",")] if only else range(world)):
        if world > 1: os.environ["CGH_EMULATE_DEVICE"] = str(d)
        ses.prove_plain(w, r, s)
        plain.append(min(ses.prove_plain(w, r, s)[1] for _ in range(3)) * 1e3)
        ses.prove_rep3(w[:2], [a, b, c], [c, a, b], streams, solo=False)
        party.append(min(ses.prove_rep3(w[:2], [a, b, c], [c, a, b], streams)[2] for _ in range(3)) * 1e3)
    os.environ.pop("CGH_EMULATE_DEVICE", None)
    ses.close()
    fmt = lambda v: " ".join(f"{x:.1f}" for x in v)
    print(f"2^{log_m}, {world} device(s){tag}: plain max {max(plain):.1f} ms (device {int(np.argmax(plain))}; per device {fmt(plain)}), "
          f"one REP3 party alone max {max(party):.1f} ms (device {int(np.argmax(party))}; per device {fmt(party)})", flush=True)
