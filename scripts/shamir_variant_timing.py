"""Three Shamir parties (threshold 1) on one GPU, each through cgh_session_prove_shamir_party, with and without the opt-in degree-2t quotient
variant (CGH_SESSION_ADDITIVE_H).  The transport is an in-memory queue mesh behind Python callbacks and the party's randomness a Python
callback over a pre-drawn stream, so the reference-protocol figure carries the cost of moving 2 x 32 B x m x (n - 1) bytes through Python;
the variant exchanges a few points and field elements only.  usage: python scripts/shamir_variant_timing.py [log_m ...]"""
import ctypes as C
import importlib, os, queue, shutil, sys, tempfile, threading, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")
CURVE, N, T = cg.BN254, 3, 1


class QueueEnd:
    def __init__(self, me, qs, stream):
        self.me, self.qs, self.stream, self.k, self.bytes = me, qs, stream, 0, 0
        self._cbs = (cg._SH_SEND(self._send), cg._SH_RECV(self._recv), cg._SH_RAND(self._rand))
        self.net = cg.ShamirNetTable(None, me, N, self._cbs[0], self._cbs[1])
        self.rand = cg.ShamirRandTable(None, self._cbs[2])

    def _send(self, u, to, data, n): self.qs[(self.me, to)].put(C.string_at(data, n)); self.bytes += n; return 0

    def _recv(self, u, frm, data, n):
        try: msg = self.qs[(frm, self.me)].get(timeout=60)
        except queue.Empty: return 110
        if len(msg) != n: return 74
        C.memmove(data, msg, n); return 0

    def _rand(self, u, n, out):
        if self.k + n > self.stream.shape[0]: return 1
        C.memmove(out, self.stream[self.k:self.k + n].ctypes.data, 32 * n); self.k += n
        return 0


def field_stream(count, seed):
    v = np.random.default_rng(seed).integers(0, 1 << 63, size=(count, 4), dtype=np.uint64)
    v[:, 3] &= np.uint64((1 << 60) - 1)                                  # < the BN254 scalar modulus: any such limbs are a field element (Montgomery form)
    return v


def main():
    ctx = cg.Context(0)
    dev = torch.device("cuda", 0)
    for log_m in [int(x) for x in sys.argv[1:]] or [20]:
        d = tempfile.mkdtemp(prefix="cg_shamir_")
        try:
            zp, wp = os.path.join(d, "s.zkey"), os.path.join(d, "s.wtns")
            cg.host_synth_circuit(CURVE, log_m, 0x5EED, zp, wp, device=0)
            w = cg.host_read_wtns(CURVE, wp)
            m, n_aux = 1 << log_m, w.shape[0] - 2
            # shares of the private witness: w + r * x at x = 1, 2, 3 (shamir_core.rs:8-31 with t = 1), built with the ABI's own additions
            dw = torch.from_numpy(np.ascontiguousarray(w[2:]).view(np.int64)).to(dev)
            dr = torch.from_numpy(field_stream(n_aux, 7).view(np.int64)).to(dev)
            wits, cur = [], dw
            for _ in range(N):
                nxt = torch.empty_like(dw); ctx.vec_add(CURVE, nxt, cur, dr, n_aux); ctx.sync(); torch.cuda.synchronize()
                wits.append(nxt.cpu().numpy().view(np.uint64)); cur = nxt
            del dw, dr, cur
            for additive in (True, False):
                # `amount` secrets give amount * (t + 1) double sharings for amount * (1 + 3t) draws; the king draws t more per re-shared element
                pre = 8 if additive else (2 * m + 8) // (T + 1) + 1
                streams = [field_stream((pre + 2048) * (1 + 3 * T) + T * (2 * m + 64) + 4096, 100 + i) for i in range(N)]
                ses = cg.ProvingSession(CURVE, zp, precompute=True, validate=False, additive_h=additive)
                times = []
                for rep in range(3):
                    qs = {(a, b): queue.Queue() for a in range(N) for b in range(N) if a != b}
                    ends = [QueueEnd(i, qs, streams[i]) for i in range(N)]
                    out, errs, secs = [None] * N, [None] * N, [0.0] * N

                    def party(i):
                        try: out[i], secs[i] = cg.host_prove_shamir_party(ses, T, w[:2], wits[i], ends[i].net, ends[i].rand, preprocess=pre)
                        except Exception as e: errs[i] = e
                    th = [threading.Thread(target=party, args=(i,)) for i in range(N)]
                    t0 = time.perf_counter()
                    for x in th: x.start()
                    for x in th: x.join()
                    dt = time.perf_counter() - t0
                    if any(errs): raise RuntimeError(errs)
                    assert all((out[i] == out[0]).all() for i in range(N)), "parties disagree"
                    times.append((dt, max(secs), ends[0].bytes))
                ses.close()
                dt, sec, sent = min(times)
                print(f"2^{log_m} Shamir 3 parties (t = 1) on one GPU, {'degree-2t quotient variant' if additive else 'reference protocol'}: three proofs in {dt * 1e3:.1f} ms wall "
                      f"(slowest party's prove call {sec * 1e3:.1f} ms, preprocess({pre}) included); party 0 sent {sent / 1e6:.3f} MB", flush=True)
        finally:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
