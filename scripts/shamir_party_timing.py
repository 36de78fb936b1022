"""Three Shamir parties (threshold 1) on ONE GPU, each through cgh_session_prove_shamir_party_seeded, joined by the library's in-memory mesh
(cgh_shamir_loopback_*: no Python in the data path), private randomness = a ChaCha12 stream per party with the preprocess batch drawn on the
GPU.  Reference protocol (degree reduction through the king after every mul_vec) and, beside it, the opt-in degree-2t quotient variant.
usage: python scripts/shamir_party_timing.py [log_m ...]"""
import importlib, os, shutil, sys, tempfile, threading, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
CURVE, N, T = cg.BN254, 3, 1


def field_stream(count, seed):
    v = np.random.default_rng(seed).integers(0, 1 << 63, size=(count, 4), dtype=np.uint64)
    v[:, 3] &= np.uint64((1 << 60) - 1)
    return v


def main():
    ctx = cg.Context(0)
    dev = torch.device("cuda", 0)
    for log_m in [int(x) for x in sys.argv[1:]] or [18, 20, 22]:
        d = tempfile.mkdtemp(prefix="cg_shamir_")
        try:
            zp, wp = os.path.join(d, "s.zkey"), os.path.join(d, "s.wtns")
            cg.host_synth_circuit(CURVE, log_m, 0x5EED, zp, wp, device=0)
            w = cg.host_read_wtns(CURVE, wp)
            m, n_aux = 1 << log_m, w.shape[0] - 2
            dw = torch.from_numpy(np.ascontiguousarray(w[2:]).view(np.int64)).to(dev)
            dr = torch.from_numpy(field_stream(n_aux, 7).view(np.int64)).to(dev)
            wits, cur = [], dw
            for _ in range(N):                                        # w + r x at x = 1, 2, 3 (shamir_core.rs:8-31, t = 1)
                nxt = torch.empty_like(dw); ctx.vec_add(CURVE, nxt, cur, dr, n_aux); ctx.sync(); torch.cuda.synchronize()
                wits.append(nxt.cpu().numpy().view(np.uint64)); cur = nxt
            del dw, dr, cur
            seeds = [bytes((31 * i + 7 * k + 1) & 255 for k in range(32)) for i in range(N)]
            for additive in (False, True):
                pre = 8 if additive else (2 * m + 8) // (T + 1) + 1
                ses = cg.ProvingSession(CURVE, zp, precompute=True, validate=False, additive_h=additive)
                times = []
                for rep in range(3):
                    hub = cg.ShamirLoopbackHub(N)
                    nets = [hub.net(i, record=(rep == 2)) for i in range(N)]
                    out, errs, secs = [None] * N, [None] * N, [0.0] * N

                    def party(i):
                        try: out[i], secs[i] = cg.host_prove_shamir_party_seeded(ses, T, w[:2], wits[i], nets[i], seeds[i], preprocess=pre)
                        except Exception as e: errs[i] = e; hub.abort()
                    th = [threading.Thread(target=party, args=(i,)) for i in range(N)]
                    t0 = time.perf_counter()
                    for x in th: x.start()
                    for x in th: x.join()
                    dt = time.perf_counter() - t0
                    if any(errs): raise RuntimeError(errs)
                    assert all((out[i] == out[0]).all() for i in range(N)), "parties disagree"
                    times.append((dt, max(secs)))
                    if rep == 2:                                       # each party ALONE on the GPU, served what it received (network excluded)
                        solo = []
                        for i in range(N):
                            best = 1e9
                            for _ in range(3):
                                got, sec = cg.host_prove_shamir_party_seeded(ses, T, w[:2], wits[i], hub.replay_net(i), seeds[i], preprocess=pre)
                                assert (got == out[0]).all(), "replayed party produced a different proof"
                                best = min(best, sec)
                            solo.append(best)
                    hub.close()
                ses.close()
                dt, ps = min(times)
                print(f"2^{log_m} Shamir 3 parties (t = 1) on one GPU, {'degree-2t quotient variant' if additive else 'reference protocol'}: three proofs in {dt * 1e3:.1f} ms wall "
                      f"(slowest party's prove call {ps * 1e3:.1f} ms, preprocess({pre}) with its {pre * (1 + 3 * T)} draws included); "
                      f"each party alone, its received messages replayed: king {solo[0] * 1e3:.1f} ms, parties 1 / 2 {solo[1] * 1e3:.1f} / {solo[2] * 1e3:.1f} ms", flush=True)
        finally:
            shutil.rmtree(d, ignore_errors=True)


main()
