"""ONE REP3 party behind the C callback ABI (cgh_session_prove_rep3_party, include/cogroth16_host.h): what `co-circom generate-proof
--protocol REP3` runs per process (co-circom/co-circom/src/bin/co-circom.rs:484-506) — the party's own shares, the caller's network
(Rep3Network, mpc-core/src/protocols/rep3/network.rs:13-64) and the caller's correlated randomness (Rep3Rand, rep3/rngs.rs:25-62).
CPU part: the transports and randomness sources that fill the callback tables.  GPU part (-m gpu): three parties on three threads,
each through the callback ABI over real sockets (length-delimited frames like mpc-net), give the oracle's proofs bit for bit."""
import ctypes as C
import os
import queue
import socket
import struct
import threading

import numpy as np
import pytest

import oracle_lib as orc
from oracle_lib import BN254, BLS12_381, FR
from product import cg, ensure_built

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fx(curve_name, circuit, f):
    return os.path.join(GOLDEN, "groth16", curve_name, circuit, f)


def rep3_share(curve, vals, rng):
    a = orc.random_field(curve, FR, vals.shape[0], rng); b = orc.random_field(curve, FR, vals.shape[0], rng)
    c = orc.field_op(curve, FR, "sub", orc.field_op(curve, FR, "sub", vals, a), b)
    return [a, b, c], [c, a, b]


# ---- a transport over stream sockets: u64 length prefix + payload per message (mpc-net's LengthDelimitedCodec framing); a reader
# thread per incoming connection queues the frames, so that three parties sending 4 MiB chunks at once cannot block each other
class SocketEnd:
    def __init__(self, party, to_next, from_prev, to_prev, from_next):
        self.party = party
        self.out = {"next": to_next, "prev": to_prev}
        self.inq = {"prev": queue.Queue(), "next": queue.Queue()}
        self.sent = {"next": 0, "prev": 0}
        self.sent_bytes = 0
        self.fail_send_next_at = None
        self.corrupt_send_next_at = None
        self.corrupt_top_byte_at = None
        self.readers = [threading.Thread(target=self._reader, args=(from_prev, self.inq["prev"]), daemon=True),
                        threading.Thread(target=self._reader, args=(from_next, self.inq["next"]), daemon=True)]
        for t in self.readers: t.start()
        self._cbs = (cg._SEND(lambda u, d, n: self._send("next", d, n)), cg._RECV(lambda u, d, n: self._recv("prev", d, n)),
                     cg._SEND(lambda u, d, n: self._send("prev", d, n)), cg._RECV(lambda u, d, n: self._recv("next", d, n)))
        self.table = cg.Rep3NetTable(None, party, self._cbs[0], self._cbs[1], self._cbs[2], self._cbs[3], cg._RECV_PINNED())

    @staticmethod
    def _reader(sock, q):
        try:
            while True:
                hdr = b""
                while len(hdr) < 8:
                    part = sock.recv(8 - len(hdr))
                    if not part: q.put(None); return
                    hdr += part
                n, = struct.unpack("<Q", hdr)
                buf = bytearray(n); view = memoryview(buf); got = 0
                while got < n:
                    k = sock.recv_into(view[got:], n - got)
                    if k == 0: q.put(None); return
                    got += k
                q.put(bytes(buf))
        except OSError:
            q.put(None)

    def _send(self, where, data, n):
        try:
            if where == "next" and self.fail_send_next_at is not None and self.sent["next"] == self.fail_send_next_at:
                return 32                                                          # EPIPE: the peer went away
            payload = C.string_at(data, n)
            if where == "next" and self.corrupt_send_next_at is not None and self.sent["next"] == self.corrupt_send_next_at:
                payload = bytes([payload[0] ^ 1]) + payload[1:]                    # a bit flipped on the wire
            if where == "next" and self.corrupt_top_byte_at is not None and self.sent["next"] == self.corrupt_top_byte_at:
                payload = payload[:31] + b"\xff" + payload[32:]                   # first element: top limb above the modulus
            self.out[where].sendall(struct.pack("<Q", n) + payload)
            self.sent[where] += 1; self.sent_bytes += n
            return 0
        except OSError as e:
            return e.errno or 5

    def _recv(self, where, data, n):
        try: msg = self.inq[where].get(timeout=120)
        except queue.Empty: return 110                                             # ETIMEDOUT
        if msg is None: return 104                                                 # ECONNRESET
        if len(msg) != n: return 74                                                # EBADMSG: rep3.rs:663-668 "invalid number of elements received"
        C.memmove(data, msg, n)
        return 0

    def close(self):
        for s in self.out.values():
            try: s.shutdown(socket.SHUT_RDWR)
            except OSError: pass
            s.close()


def socket_ring():
    """three SocketEnds joined pairwise by socketpairs: i -> i+1 (next direction) and i -> i-1 (prev direction)"""
    fwd = [socket.socketpair() for _ in range(3)]          # fwd[i]: party i writes [0], party i+1 reads [1]
    bwd = [socket.socketpair() for _ in range(3)]          # bwd[i]: party i writes [0], party i-1 reads [1]
    return [SocketEnd(i, fwd[i][0], fwd[(i + 2) % 3][1], bwd[i][0], bwd[(i + 1) % 3][1]) for i in range(3)]


class PyRand:
    """Rep3Rand written in Python over two streams: fills the library's buffer (the other way of answering masking_field_elements),
    G * a - G * b as masking point (the oracle's stand-in for C::rand, oracle/groth16.hpp)"""

    def __init__(self, curve, rng1, rng2):
        self.curve, self.r1, self.r2, self.k = curve, rng1, rng2, 0
        self.diff = orc.field_op(curve, FR, "sub", rng1, rng2)
        self._cbs = (cg._MASKS(self._masks), cg._FES(self._fes), cg._EC(self._ec))
        self.table = cg.Rep3RandTable(None, *self._cbs)

    def _masks(self, u, n, buf, out):
        if self.k + n > self.diff.shape[0]: return 1
        C.memmove(buf, self.diff[self.k:self.k + n].ctypes.data, 32 * n); out[0] = buf; self.k += n
        return 0

    def _fes(self, u, a, b):
        C.memmove(a, self.r1[self.k].ctypes.data, 32); C.memmove(b, self.r2[self.k].ctypes.data, 32); self.k += 1
        return 0

    def _ec(self, u, group, out):
        g = cg.point_generator(self.curve, group)
        m = cg.point_add(self.curve, group, cg.point_scalar_mul(self.curve, group, g, self.r1[self.k]),
                         cg.point_neg(self.curve, group, cg.point_scalar_mul(self.curve, group, g, self.r2[self.k])))
        self.k += 1
        m = np.ascontiguousarray(m, dtype=np.uint64); C.memmove(out, m.ctypes.data, m.nbytes)
        return 0


def run_three_parties(ses, pub, wa, wb, nets, rands):
    out, errs = [None] * 3, [None] * 3

    def party(i):
        try: out[i], _ = cg.host_prove_rep3_party(ses, pub, wa[i], wb[i], nets[i], rands[i])
        except Exception as e: errs[i] = e
    th = [threading.Thread(target=party, args=(i,)) for i in range(3)]
    for t in th: t.start()
    for t in th: t.join(300)
    return out, errs


# ---------------------------------------------------------------------------------------------------------------- CPU
def test_loopback_tables_move_messages_between_threads():
    ensure_built()
    hub = cg.LoopbackHub()
    try:
        nets = [hub.net(i, record=(i == 1)) for i in range(3)]
        assert [n.party_id for n in nets] == [0, 1, 2]
        msg = np.arange(1000, dtype=np.uint64)
        got = np.zeros_like(msg); back = np.zeros(4, dtype=np.uint64)
        assert nets[0].send_next(nets[0].user, msg.ctypes.data, msg.nbytes) == 0          # 0 -> 1
        assert nets[2].send_prev(nets[2].user, msg[:4].ctypes.data, 32) == 0              # 2 -> 1 (prev direction)
        t = threading.Thread(target=lambda: (nets[1].recv_prev(nets[1].user, got.ctypes.data, got.nbytes), nets[1].recv_next(nets[1].user, back.ctypes.data, 32)))
        t.start(); t.join(30)
        np.testing.assert_array_equal(got, msg); np.testing.assert_array_equal(back, msg[:4])
        # a message of the wrong size is an error (rep3.rs:663-668), not a truncated read
        assert nets[1].send_next(nets[1].user, msg.ctypes.data, 64) == 0
        assert nets[2].recv_prev(nets[2].user, got.ctypes.data, 32) != 0
        assert b"invalid number of bytes" in cg.load_host().cgh_last_error()
        # party 1's traffic was recorded: the replay table serves it again, sends are dropped
        rp = hub.replay_net(1)
        got[:] = 0; back[:] = 0
        assert rp.send_next(rp.user, msg.ctypes.data, 8) == 0
        assert rp.recv_prev(rp.user, got.ctypes.data, got.nbytes) == 0 and rp.recv_next(rp.user, back.ctypes.data, 32) == 0
        np.testing.assert_array_equal(got, msg); np.testing.assert_array_equal(back, msg[:4])
        assert rp.recv_prev(rp.user, got.ctypes.data, 8) != 0                             # nothing left
        # abort wakes a waiting receiver
        res = []
        t = threading.Thread(target=lambda: res.append(nets[0].recv_prev(nets[0].user, got.ctypes.data, 8)))
        t.start(); hub.abort(); t.join(30)
        assert res and res[0] != 0
    finally:
        hub.close()


def test_shamir_loopback_mesh_moves_messages_any_to_any():
    """cgh_shamir_loopback_*: n parties of one process behind cgh_shamir_net tables (shamir/network.rs:17-59: one message = one send / recv
    pair between two parties, FIFO per directed pair); a wrong size is an error, abort wakes a waiting receiver"""
    ensure_built()
    with pytest.raises(cg.BackendError):
        cg.ShamirLoopbackHub(2)                                                           # shamir/network.rs:75-77: at least three parties
    hub = cg.ShamirLoopbackHub(4)
    try:
        with pytest.raises(cg.BackendError):
            hub.net(4)
        nets = [hub.net(i, record=(i == 3)) for i in range(4)]
        assert [(n.party_id, n.num_parties) for n in nets] == [(0, 4), (1, 4), (2, 4), (3, 4)]
        with pytest.raises(cg.BackendError):
            hub.replay_net(1)                                                             # nothing was recorded for party 1
        a = np.arange(500, dtype=np.uint64); b = np.arange(7, dtype=np.uint64) + 1000
        assert nets[0].send(nets[0].user, 3, a.ctypes.data, a.nbytes) == 0
        assert nets[0].send(nets[0].user, 3, b.ctypes.data, b.nbytes) == 0                # second message on the same pair: FIFO
        assert nets[2].send(nets[2].user, 1, b.ctypes.data, b.nbytes) == 0
        ga = np.zeros_like(a); gb = np.zeros_like(b); gc = np.zeros_like(b)
        t = threading.Thread(target=lambda: (nets[3].recv(nets[3].user, 0, ga.ctypes.data, ga.nbytes), nets[3].recv(nets[3].user, 0, gb.ctypes.data, gb.nbytes),
                                             nets[1].recv(nets[1].user, 2, gc.ctypes.data, gc.nbytes)))
        t.start(); t.join(30)
        np.testing.assert_array_equal(ga, a); np.testing.assert_array_equal(gb, b); np.testing.assert_array_equal(gc, b)
        # party 3's traffic was recorded: the replay table serves it again (twice: every table starts from the first message), sends are dropped
        for _ in range(2):
            rp = hub.replay_net(3)
            ga[:] = 0; gb[:] = 0
            assert rp.send(rp.user, 0, a.ctypes.data, 8) == 0
            assert rp.recv(rp.user, 0, ga.ctypes.data, ga.nbytes) == 0 and rp.recv(rp.user, 0, gb.ctypes.data, gb.nbytes) == 0
            np.testing.assert_array_equal(ga, a); np.testing.assert_array_equal(gb, b)
            assert rp.recv(rp.user, 0, ga.ctypes.data, 8) != 0 and rp.recv(rp.user, 1, ga.ctypes.data, 8) != 0      # nothing left / nothing from party 1
        assert nets[1].send(nets[1].user, 0, a.ctypes.data, 64) == 0
        assert nets[0].recv(nets[0].user, 1, ga.ctypes.data, 32) != 0                      # shamir.rs:324-329
        assert b"Invalid number of elements" in cg.load_host().cgh_last_error()
        res = []
        t = threading.Thread(target=lambda: res.append(nets[2].recv(nets[2].user, 3, ga.ctypes.data, 8)))
        t.start(); hub.abort(); t.join(30)
        assert res and res[0] != 0
    finally:
        hub.close()


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
def test_stream_rand_follows_rep3rand(curve):
    """rngs.rs:37-46: masking element = rand(rng1) - rand(rng2), random_fes = the pair; every draw advances both streams by one"""
    ensure_built()
    rng = np.random.default_rng(4)
    n = 5000
    r1 = orc.random_field(curve, FR, n, rng); r2 = orc.random_field(curve, FR, n, rng)
    r2[7] = r1[7]; r1[8] = 0; r2[9] = 0                                                  # difference 0, minuend 0, subtrahend 0
    src = cg.StreamRand(curve, r1, r2)
    try:
        t = src.table
        out = C.c_void_p()
        assert t.masking_field_elements(t.user, 4000, None, C.byref(out)) == 0
        got = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(4000, 4))
        np.testing.assert_array_equal(got, orc.field_op(curve, FR, "sub", r1[:4000], r2[:4000]))
        a = np.zeros(4, dtype=np.uint64); b = np.zeros(4, dtype=np.uint64)
        assert t.random_fes(t.user, a.ctypes.data, b.ctypes.data) == 0
        np.testing.assert_array_equal(a, r1[4000]); np.testing.assert_array_equal(b, r2[4000])
        assert t.masking_field_elements(t.user, 999, None, C.byref(out)) == 0
        got = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(999, 4))
        np.testing.assert_array_equal(got, orc.field_op(curve, FR, "sub", r1[4001:5000], r2[4001:5000]))
        assert t.random_fes(t.user, a.ctypes.data, b.ctypes.data) != 0                    # exhausted
        assert b"exhausted" in cg.load_host().cgh_last_error()
    finally:
        src.close()


def test_party_entries_reject_bad_arguments_without_a_gpu():
    """the one-party entries validate what they are handed before touching a device: null pointers, a party id outside 0..2, missing
    required callbacks, fewer than three Shamir parties (shamir/network.rs:75-77) — status 1 and a message, never a crash"""
    ensure_built()
    h = cg.load_host()
    out = np.zeros(32, dtype=np.uint64); pub = np.zeros((2, 4), dtype=np.uint64)
    net = cg.Rep3NetTable(); rnd = cg.Rep3RandTable()
    assert h.cgh_session_prove_rep3_party(None, None, None, None, C.byref(net), C.byref(rnd), None, None) != 0
    assert b"null argument" in h.cgh_last_error()
    sn = cg.ShamirNetTable(); sr = cg.ShamirRandTable()
    assert h.cgh_session_prove_shamir_party(None, 1, None, None, C.byref(sn), C.byref(sr), C.c_size_t(0), None, None) != 0
    assert b"null argument" in h.cgh_last_error()
    # loopback tables for a party id that does not exist
    hub = cg.LoopbackHub()
    try:
        with pytest.raises(cg.BackendError):
            hub.net(3)
        with pytest.raises(cg.BackendError):
            hub.replay_net(-1)
    finally:
        hub.close()


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_three_parties_over_sockets_give_the_oracle_proof(curve_name):
    """tests/tests/circom/e2e_tests/mod.rs:33-82, one thread per party, each through the callback ABI over sockets (poseidon fixture)"""
    ensure_built()
    curve = {"bn254": BN254, "bls12_381": BLS12_381}[curve_name]
    zpath = fx(curve_name, "poseidon", "circuit.zkey")
    z = orc.ZKey(curve, zpath); w = orc.read_wtns(curve, fx(curve_name, "poseidon", "witness.wtns"))
    rng = np.random.default_rng(15)
    pub = w[:z.n_public + 1]
    wa, wb = rep3_share(curve, w[z.n_public + 1:], rng)
    streams = [orc.random_field(curve, FR, 2 * z.domain_size + 4, rng) for _ in range(3)]
    want = z.prove_rep3(pub, wa, wb, streams)
    ses = cg.ProvingSession(curve, zpath, precompute=False)
    ends = socket_ring()
    rands = [PyRand(curve, streams[0], streams[2]), cg.StreamRand(curve, streams[1], streams[0]), cg.StreamRand(curve, streams[2], streams[1])]
    try:
        out, errs = run_three_parties(ses, pub, wa, wb, [e.table for e in ends], [r.table for r in rands])
        assert errs == [None, None, None], errs
        np.testing.assert_array_equal(np.stack(out), want)
        vk = orc.vk_from_json(curve, fx(curve_name, "poseidon", "verification_key.json"))
        assert orc.verify(curve, vk, w[1:1 + z.n_public], out[0])
        assert all(e.sent["next"] > 0 for e in ends)                                      # the proofs really crossed the sockets
    finally:
        for e in ends: e.close()
        for r in rands[1:]: r.close()
        ses.close()


@pytest.mark.gpu
@pytest.mark.parametrize("chunked", [False, True])
def test_three_parties_over_sockets_at_2_16(chunked, tmp_path, host_option):
    """BASELINE configs[1] scale.  chunked: the two mul_vec exchanges travel as asynchronous 128 KiB chunks (the path a 2^22 proof
    takes with 4 MiB chunks), forced here by lowering the threshold"""
    ensure_built()
    if chunked: host_option(cg.HOST_OPT_XCHG_ASYNC_MIN, 4096)
    curve, log_m = BN254, 16
    threads = min(32, os.cpu_count() or 8)
    zp, wp = str(tmp_path / "s.zkey"), str(tmp_path / "s.wtns")
    orc.make_synthetic(curve, log_m, 31, zp, wp, threads=threads)
    z = orc.ZKey(curve, zp); w = orc.read_wtns(curve, wp)
    rng = np.random.default_rng(19)
    wa, wb = rep3_share(curve, w[2:], rng)
    streams = [orc.random_field(curve, FR, 2 * z.domain_size + 4, rng) for _ in range(3)]
    want = z.prove_rep3(w[:2], wa, wb, streams, threads=threads)
    ses = cg.ProvingSession(curve, zp, precompute=True)
    ends = socket_ring()
    rands = [cg.StreamRand(curve, streams[i], streams[(i + 2) % 3]) for i in range(3)]
    try:
        out, errs = run_three_parties(ses, w[:2], wa, wb, [e.table for e in ends], [r.table for r in rands])
        assert errs == [None, None, None], errs
        np.testing.assert_array_equal(np.stack(out), want)
        if chunked: assert ends[0].sent["next"] > 2 * 4                                   # 2 exchanges in several chunks + the O(1) rounds
        # the same session through the loopback transport, then party 0 alone on its recorded traffic
        hub = cg.LoopbackHub()
        r2 = [cg.StreamRand(curve, streams[i], streams[(i + 2) % 3]) for i in range(3)] + [cg.StreamRand(curve, streams[0], streams[2])]
        try:
            out2, errs = run_three_parties(ses, w[:2], wa, wb, [hub.net(i, record=(i == 0)) for i in range(3)], [r.table for r in r2[:3]])
            assert errs == [None, None, None], errs
            np.testing.assert_array_equal(np.stack(out2), want)
            solo, sec = cg.host_prove_rep3_party(ses, w[:2], wa[0], wb[0], hub.replay_net(0), r2[3].table)
            np.testing.assert_array_equal(solo, want[0]); assert sec > 0
        finally:
            for r in r2: r.close()
            hub.close()
    finally:
        for e in ends: e.close()
        for r in rands: r.close()
        ses.close()


@pytest.mark.gpu
def test_a_failing_network_callback_fails_the_proof():
    """std::io::Error from the network ends the prove (rep3.rs:661-669 `?`): the code reaches the caller, nothing hangs, the session
    stays usable"""
    ensure_built()
    curve = BN254
    zpath = fx("bn254", "poseidon", "circuit.zkey")
    z = orc.ZKey(curve, zpath); w = orc.read_wtns(curve, fx("bn254", "poseidon", "witness.wtns"))
    rng = np.random.default_rng(23)
    pub = w[:z.n_public + 1]
    wa, wb = rep3_share(curve, w[z.n_public + 1:], rng)
    streams = [orc.random_field(curve, FR, 2 * z.domain_size + 4, rng) for _ in range(3)]
    ses = cg.ProvingSession(curve, zpath, precompute=False)
    try:
        ends = socket_ring()
        ends[1].fail_send_next_at = 1                                                     # party 1's second message never leaves
        rands = [cg.StreamRand(curve, streams[i], streams[(i + 2) % 3]) for i in range(3)]
        out, errs = [None] * 3, [None] * 3

        def party(i):
            try: out[i], _ = cg.host_prove_rep3_party(ses, pub, wa[i], wb[i], ends[i].table, rands[i].table)
            except Exception as e: errs[i] = e
            finally: ends[i].close()                                                      # a dying process closes its connections
        th = [threading.Thread(target=party, args=(i,)) for i in range(3)]
        for t in th: t.start()
        for t in th: t.join(120)
        assert not any(t.is_alive() for t in th)
        assert errs[1] is not None and "send_next failed with code 32" in str(errs[1])
        assert errs[2] is not None and "recv_prev failed" in str(errs[2])
        for r in rands: r.close()
        # the session still proves
        ends = socket_ring()
        rands = [cg.StreamRand(curve, streams[i], streams[(i + 2) % 3]) for i in range(3)]
        out, errs = run_three_parties(ses, pub, wa, wb, [e.table for e in ends], [r.table for r in rands])
        assert errs == [None, None, None], errs
        np.testing.assert_array_equal(np.stack(out), z.prove_rep3(pub, wa, wb, streams))
        for e in ends: e.close()
        for r in rands: r.close()
    finally:
        ses.close()


# ---------------------------------------------------------------------------------------------------------------- Shamir twin
class ShamirMeshEnd:
    """party `me` of n: one socket per directed pair, framed messages, a reader thread per incoming socket (as SocketEnd)"""

    def __init__(self, me, n, outs, ins, stream):
        self.me, self.n, self.out = me, n, outs
        self.inq = {p: queue.Queue() for p in ins}
        self.readers = [threading.Thread(target=SocketEnd._reader, args=(sock, self.inq[p]), daemon=True) for p, sock in ins.items()]
        for r in self.readers: r.start()
        self.stream, self.k, self.sent = stream, 0, 0
        self._cbs = (cg._SH_SEND(self._send), cg._SH_RECV(self._recv), cg._SH_RAND(self._rand))
        self.net = cg.ShamirNetTable(None, me, n, self._cbs[0], self._cbs[1])
        self.rand = cg.ShamirRandTable(None, self._cbs[2])

    def _send(self, u, to, data, nbytes):
        try: self.out[to].sendall(struct.pack("<Q", nbytes) + C.string_at(data, nbytes)); self.sent += 1; return 0
        except OSError as e: return e.errno or 5

    def _recv(self, u, frm, data, nbytes):
        try: msg = self.inq[frm].get(timeout=120)
        except queue.Empty: return 110
        if msg is None: return 104
        if len(msg) != nbytes: return 74
        C.memmove(data, msg, nbytes)
        return 0

    def _rand(self, u, n, out):
        if self.k + n > self.stream.shape[0]: return 1
        C.memmove(out, self.stream[self.k:self.k + n].ctypes.data, 32 * n); self.k += n
        return 0

    def close(self):
        for s in self.out.values():
            try: s.shutdown(socket.SHUT_RDWR)
            except OSError: pass
            s.close()


def shamir_mesh(n, streams):
    pairs = {(a, b): socket.socketpair() for a in range(n) for b in range(n) if a != b}      # (a, b): a writes [0], b reads [1]
    return [ShamirMeshEnd(i, n, {b: pairs[(i, b)][0] for b in range(n) if b != i}, {a: pairs[(a, i)][1] for a in range(n) if a != i}, streams[i]) for i in range(n)]


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,n,t,preprocess", [("bn254", 3, 1, 0), ("bn254", 5, 2, 600), ("bls12_381", 3, 1, 0)])
def test_shamir_parties_over_sockets_give_the_oracle_proof(curve_name, n, t, preprocess):
    """cgh_session_prove_shamir_party: n threads, each ONE party through the C callback ABI (any-to-any sockets, the party's private
    randomness served by a callback), poseidon fixture; lazy double sharings (batches of 1024) and preprocess(amount) on the GPU"""
    from test_shamir import setup
    ensure_built()
    curve, z, w, wits, streams = setup(curve_name, "poseidon", n, t, seed=41, preprocess=preprocess)
    pub = w[:z.n_public + 1]
    want = orc.prove_shamir(z, n, t, pub, wits, streams, preprocess=preprocess)
    ses = cg.ProvingSession(curve, fx(curve_name, "poseidon", "circuit.zkey"), precompute=False)
    ends = shamir_mesh(n, streams)
    out, errs = [None] * n, [None] * n

    def party(i):
        try: out[i], _ = cg.host_prove_shamir_party(ses, t, pub, wits[i], ends[i].net, ends[i].rand, preprocess=preprocess)
        except Exception as e: errs[i] = e
    try:
        th = [threading.Thread(target=party, args=(i,)) for i in range(n)]
        for x in th: x.start()
        for x in th: x.join(300)
        assert errs == [None] * n, errs
        np.testing.assert_array_equal(np.stack(out), want)
        assert all(e.sent > 0 for e in ends)
    finally:
        for e in ends: e.close()
        ses.close()


@pytest.mark.variant
@pytest.mark.parametrize("curve_name,circuit", [("bn254", "poseidon"), ("bls12_381", "multiplier2")])
def test_additive_quotient_variant_gives_the_same_proofs(curve_name, circuit):
    """CGH_SESSION_ADDITIVE_H (opt-in, not the reference's message sequence): the witness map's products stay masked local products, MSMs run
    on the own component, five points are re-shared in one round — the three proofs are the ORACLE's (reference protocol) bit for bit on
    the same shares and randomness, and the parties exchange O(1) bytes instead of 2 x 32 B x domain_size"""
    ensure_built()
    curve = {"bn254": BN254, "bls12_381": BLS12_381}[curve_name]
    zpath = fx(curve_name, circuit, "circuit.zkey")
    z = orc.ZKey(curve, zpath); w = orc.read_wtns(curve, fx(curve_name, circuit, "witness.wtns"))
    rng = np.random.default_rng(615)
    pub = w[:z.n_public + 1]
    wa, wb = rep3_share(curve, w[z.n_public + 1:], rng)
    streams = [orc.random_field(curve, FR, 2 * z.domain_size + 4, rng) for _ in range(3)]
    want = z.prove_rep3(pub, wa, wb, streams)
    sent = {}
    for additive in (False, True):
        ses = cg.ProvingSession(curve, zpath, precompute=False, additive_h=additive)
        ends = socket_ring()
        rands = [cg.StreamRand(curve, streams[i], streams[(i + 2) % 3]) for i in range(3)]
        try:
            out, errs = run_three_parties(ses, pub, wa, wb, [e.table for e in ends], [r.table for r in rands])
            assert errs == [None, None, None], errs
            np.testing.assert_array_equal(np.stack(out), want)
            sent[additive] = ends[0].sent_bytes
        finally:
            for e in ends: e.close()
            for r in rands: r.close()
            ses.close()
    nq = 48 if curve == BLS12_381 else 32
    assert sent[False] - sent[True] == 2 * 32 * z.domain_size - (4 * 2 * nq + 4 * nq)    # two vector messages fewer, one message of 4 G1 + 1 G2 more


@pytest.mark.variant
def test_additive_quotient_variant_at_2_16(tmp_path):
    """the variant on the chunked-exchange sizes (precomputed tables, second context, prefetched masks) and on a two-device session"""
    ensure_built()
    curve, log_m = BN254, 16
    threads = min(32, os.cpu_count() or 8)
    zp, wp = str(tmp_path / "s.zkey"), str(tmp_path / "s.wtns")
    orc.make_synthetic(curve, log_m, 33, zp, wp, threads=threads)
    z = orc.ZKey(curve, zp); w = orc.read_wtns(curve, wp)
    rng = np.random.default_rng(29)
    wa, wb = rep3_share(curve, w[2:], rng)
    streams = [orc.random_field(curve, FR, 2 * z.domain_size + 4, rng) for _ in range(3)]
    want = z.prove_rep3(w[:2], wa, wb, streams, threads=threads)
    for devices in (None, [0, 0]):
        ses = cg.ProvingSession(curve, zp, precompute=True, additive_h=True, devices=devices, shared_devices=True)
        hub = cg.LoopbackHub()
        rands = [cg.StreamRand(curve, streams[i], streams[(i + 2) % 3]) for i in range(3)]
        try:
            out, errs = run_three_parties(ses, w[:2], wa, wb, [hub.net(i) for i in range(3)], [r.table for r in rands])
            assert errs == [None, None, None], errs
            np.testing.assert_array_equal(np.stack(out), want)
        finally:
            for r in rands: r.close()
            hub.close(); ses.close()


@pytest.mark.variant
@pytest.mark.parametrize("curve_name,n,t", [("bn254", 3, 1), ("bn254", 5, 2), ("bls12_381", 3, 1)])
def test_shamir_degree_2t_quotient_variant(curve_name, n, t):
    """CGH_SESSION_ADDITIVE_H with Shamir parties: the witness map's products stay degree-2t sharings (no vector degree reduction, so a
    handful of double sharings instead of 2 x domain_size), h is reduced as one point after its MSM.  The blinding r, s are then other
    pairs of the same randomness than in the reference's order, so the proof is not the oracle's bit for bit: every party must hold
    the SAME proof, and it must pass the snarkjs-pinned verifier for the circuit's public inputs"""
    from test_shamir import setup
    ensure_built()
    curve, z, w, wits, streams = setup(curve_name, "poseidon", n, t, seed=43, preprocess=0)
    pub = w[:z.n_public + 1]
    vk = orc.vk_from_json(curve, fx(curve_name, "poseidon", "verification_key.json"))
    sent = {}
    for additive in (False, True):
        ses = cg.ProvingSession(curve, fx(curve_name, "poseidon", "circuit.zkey"), precompute=False, additive_h=additive)
        ends = shamir_mesh(n, streams)
        out, errs = [None] * n, [None] * n

        def party(i):
            try: out[i], _ = cg.host_prove_shamir_party(ses, t, pub, wits[i], ends[i].net, ends[i].rand, preprocess=0)
            except Exception as e: errs[i] = e
        try:
            th = [threading.Thread(target=party, args=(i,)) for i in range(n)]
            for x in th: x.start()
            for x in th: x.join(300)
            assert errs == [None] * n, errs
            for i in range(1, n): np.testing.assert_array_equal(out[i], out[0])
            assert orc.verify(curve, vk, w[1:1 + z.n_public], out[0])
            sent[additive] = sum(e.sent for e in ends)
        finally:
            for e in ends: e.close()
            ses.close()
    assert sent[True] < sent[False]


@pytest.mark.gpu
def test_a_corrupted_point_from_a_peer_is_invalid_data():
    """the reference deserialises what it receives with validation (ark-serialize behind mpc-net's recv): a point that is not on the curve
    ends the prove with InvalidData.  Party 1's open_point message (its fourth to the next party: two mul_vec vectors, r*s, the point)
    gets one bit flipped; party 2 must refuse it"""
    ensure_built()
    curve = BN254
    zpath = fx("bn254", "poseidon", "circuit.zkey")
    z = orc.ZKey(curve, zpath); w = orc.read_wtns(curve, fx("bn254", "poseidon", "witness.wtns"))
    rng = np.random.default_rng(31)
    pub = w[:z.n_public + 1]
    wa, wb = rep3_share(curve, w[z.n_public + 1:], rng)
    streams = [orc.random_field(curve, FR, 2 * z.domain_size + 4, rng) for _ in range(3)]
    ses = cg.ProvingSession(curve, zpath, precompute=False)
    try:
        ends = socket_ring()
        ends[1].corrupt_send_next_at = 3
        rands = [cg.StreamRand(curve, streams[i], streams[(i + 2) % 3]) for i in range(3)]
        out, errs = [None] * 3, [None] * 3

        def party(i):
            try: out[i], _ = cg.host_prove_rep3_party(ses, pub, wa[i], wb[i], ends[i].table, rands[i].table)
            except Exception as e: errs[i] = e
            finally: ends[i].close()
        th = [threading.Thread(target=party, args=(i,)) for i in range(3)]
        for t in th: t.start()
        for t in th: t.join(120)
        assert not any(t.is_alive() for t in th)
        assert errs[2] is not None and "invalid data" in str(errs[2]), errs
        for r in rands: r.close()
    finally:
        ses.close()


@pytest.mark.gpu
@pytest.mark.parametrize("chunked", [False, True])
def test_a_non_canonical_element_in_a_mul_vec_message_is_invalid_data(chunked, tmp_path, host_option):
    """the m-element messages of mul_vec are range-checked too — on the host when they are short single messages, on the device (behind the
    upload, read at the end of the prove) from 2^12 elements on and when they travel in chunks: an element whose limbs are not below the
    modulus = InvalidData"""
    ensure_built()
    if chunked: host_option(cg.HOST_OPT_XCHG_ASYNC_MIN, 4096)
    curve, log_m = BN254, 14
    zp, wp = str(tmp_path / "s.zkey"), str(tmp_path / "s.wtns")
    orc.make_synthetic(curve, log_m, 35, zp, wp, threads=min(32, os.cpu_count() or 8))
    z = orc.ZKey(curve, zp); w = orc.read_wtns(curve, wp)
    rng = np.random.default_rng(37)
    wa, wb = rep3_share(curve, w[2:], rng)
    streams = [orc.random_field(curve, FR, 2 * z.domain_size + 4, rng) for _ in range(3)]
    ses = cg.ProvingSession(curve, zp, precompute=False)
    try:
        ends = socket_ring()
        ends[0].corrupt_top_byte_at = 0                                                   # party 0's first mul_vec message (or its first chunk)
        rands = [cg.StreamRand(curve, streams[i], streams[(i + 2) % 3]) for i in range(3)]
        out, errs = [None] * 3, [None] * 3

        def party(i):
            try: out[i], _ = cg.host_prove_rep3_party(ses, w[:2], wa[i], wb[i], ends[i].table, rands[i].table)
            except Exception as e: errs[i] = e
            finally: ends[i].close()
        th = [threading.Thread(target=party, args=(i,)) for i in range(3)]
        for t in th: t.start()
        for t in th: t.join(120)
        assert not any(t.is_alive() for t in th)
        assert errs[1] is not None and "invalid data" in str(errs[1]), errs
        for r in rands: r.close()
    finally:
        ses.close()
