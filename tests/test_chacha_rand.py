"""Rep3Rand's draws (mpc-core/src/protocols/rep3/rngs.rs:25-46: `F::rand(&mut rng1) - F::rand(&mut rng2)`, RngType = ChaCha12Rng,
mpc-core/src/lib.rs:10) in three places that must agree: the oracle's restatement (oracle/rngs.hpp), the host library's
(collaborative-circom_amd/host/chacha.hpp, the O(1) draws and short vectors) and the device kernels (csrc/chacha_rand.hip, the m-element
masking vectors of mul_vec, rep3.rs:656-660).  The block function is pinned to the published ChaCha known-answer vectors; the draw order
(rand_chacha's word stream + ark-ff's rejection sampling) is restated from the crates' published algorithms — no reference test holds a
drawn value (PARITY UNPINNED for that part, see oracle/rngs.hpp)."""
import os
import threading

import numpy as np
import pytest

import oracle_lib as orc
from oracle_lib import BN254, BLS12_381, FR
from product import cg, ensure_built

MOD = {BN254: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
       BLS12_381: 52435875175126190479447740508185965837690552500527637822603658699938581184513}


def to_int(limbs):
    return sum(int(x) << (64 * i) for i, x in enumerate(limbs))


# ---------------------------------------------------------------------------------------------------------------- CPU
def test_chacha_block_known_answers():
    zero = np.zeros(8, dtype=np.uint32)
    # ChaCha20, all-zero key and nonce, block 0 (the classic vector; RFC 7539 A.1 #1)
    assert orc.chacha_block(20, zero, 0).tobytes().hex() == (
        "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    # RFC 7539 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00 (state words 13, 14 here: counter high / stream low)
    key = np.frombuffer(bytes(range(32)), dtype=np.uint32)
    assert orc.chacha_block(20, key, 1 | (0x09000000 << 32), 0x4a000000).tobytes().hex() == (
        "10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4ed2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    # ChaCha12 (the round count of rand_chacha::ChaCha12Rng), all-zero key and nonce, block 0 (draft-strombergson-chacha-test-vectors TC1)
    assert orc.chacha_block(12, zero, 0).tobytes().hex() == (
        "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be")


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
def test_oracle_draws_follow_fp_rand(curve):
    """ark-ff 0.4.2 `Distribution<Fp> for Standard`: four next_u64 per attempt (8 stream words), top bits cleared, below the modulus;
    a stream continued from the reported position repeats the tail of a longer draw"""
    seed = bytes((7 * i + 3) & 255 for i in range(32))
    bits = 254 if curve == BN254 else 255
    r, after = orc.chacha12_fr_rand(curve, seed, 0, 4000)
    assert all(to_int(x) < MOD[curve] for x in r)
    assert after % 8 == 0 and after >= 8 * 4000
    # re-derive the accepted values from the raw blocks: word stream = successive 12-round blocks, little-endian words
    key = np.frombuffer(seed, dtype=np.uint32)
    words = np.concatenate([orc.chacha_block(12, key, b) for b in range(after // 16 + 1)])[:after]
    cands = words.reshape(-1, 8)
    vals = [sum(int(w) << (32 * i) for i, w in enumerate(c)) & ((1 << bits) - 1) for c in cands]
    acc = [v for v in vals if v < MOD[curve]]
    assert len(acc) == 4000 and vals[-1] < MOD[curve]
    assert acc == [to_int(x) for x in r]
    assert abs(len(acc) / len(vals) - MOD[curve] / 2.0 ** bits) < 0.03
    head, mid = orc.chacha12_fr_rand(curve, seed, 0, 1500)
    tail, end = orc.chacha12_fr_rand(curve, seed, mid, 2500)
    np.testing.assert_array_equal(np.concatenate([head, tail]), r); assert end == after
    # unaligned positions (a 32-bit draw in between, e.g. the bool of C::rand) address the same word stream
    r3, a3 = orc.chacha12_fr_rand(curve, seed, 13, 50)
    words = np.concatenate([orc.chacha_block(12, key, b) for b in range(a3 // 16 + 2)])
    vals = [sum(int(w) << (32 * i) for i, w in enumerate(words[13 + 8 * k:21 + 8 * k])) & ((1 << bits) - 1) for k in range((a3 - 13) // 8)]
    assert [v for v in vals if v < MOD[curve]] == [to_int(x) for x in r3]


def test_committed_regression_vectors():
    """tests/golden/chacha_kats.json (written by make_chacha_kats.py from the oracle: regression vectors, not reference-produced): the oracle
    and the host library still draw these values and end at these positions"""
    import json
    ensure_built()
    kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chacha_kats.json")))
    assert len(kats["cases"]) == 8
    for c in kats["cases"]:
        curve = {"bn254": BN254, "bls12_381": BLS12_381}[c["curve"]]
        want = np.array([[int(l, 16) for l in v] for v in c["draws_montgomery_limbs_le"]], dtype=np.uint64)
        for fn in (orc.chacha12_fr_rand, cg.chacha12_fr_rand_host):
            got, after = fn(curve, bytes.fromhex(c["seed"]), c["word_pos"], c["n"])
            np.testing.assert_array_equal(got, want); assert after == c["word_pos_after"]


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
def test_host_library_draws_equal_the_oracle(curve):
    ensure_built()
    rng = np.random.default_rng(21 + curve)
    for pos in (0, 8, 5, 63, 64, 2**32 - 8, 2**36 + 3):
        seed = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        want, wa = orc.chacha12_fr_rand(curve, seed, pos, 777)
        got, ga = cg.chacha12_fr_rand_host(curve, seed, pos, 777)
        np.testing.assert_array_equal(got, want); assert ga == wa


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
def test_chacha_source_follows_rep3rand(curve):
    """cgh_chacha_rand_create: masking elements = rand(rng1) - rand(rng2) (rngs.rs:37-46), random_fes the pair (:42-46); the generator
    description reports and moves both positions"""
    import ctypes as C
    ensure_built()
    s1, s2 = bytes(range(32)), bytes(range(100, 132))
    src = cg.ChaChaRand(curve, s1, s2)
    try:
        t = src.table
        buf = np.zeros((1000, 4), dtype=np.uint64); out = C.c_void_p()
        assert t.masking_field_elements(t.user, 1000, buf.ctypes.data, C.byref(out)) == 0
        want, p1, p2 = orc.rep3_masks_chacha12(curve, s1, 0, s2, 0, 1000)
        np.testing.assert_array_equal(buf, want); assert src.positions() == (p1, p2)
        a = np.zeros(4, dtype=np.uint64); b = np.zeros(4, dtype=np.uint64)
        assert t.random_fes(t.user, a.ctypes.data, b.ctypes.data) == 0
        wa, q1 = orc.chacha12_fr_rand(curve, s1, p1, 1); wb, q2 = orc.chacha12_fr_rand(curve, s2, p2, 1)
        np.testing.assert_array_equal(a, wa[0]); np.testing.assert_array_equal(b, wb[0]); assert src.positions() == (q1, q2)
        st = src.streams
        g1 = (C.c_uint8 * 32)(); g2 = (C.c_uint8 * 32)(); w1 = C.c_uint64(); w2 = C.c_uint64()
        assert st.get_state(st.user, g1, C.byref(w1), g2, C.byref(w2)) == 0
        assert bytes(g1) == s1 and bytes(g2) == s2 and (w1.value, w2.value) == (q1, q2)
        assert st.set_word_pos(st.user, 160, 320) == 0 and src.positions() == (160, 320)
    finally:
        src.close()


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("curve", [BN254, BLS12_381])
def test_device_draws_equal_the_oracle(curve):
    ensure_built()
    ctx = cg.Context()
    rng = np.random.default_rng(5 + curve)
    try:
        for n, pos in ((1, 0), (2, 8), (255, 3), (1000, 13), (4097, 63), (70001, 64), (30000, 2**32 - 8), (12345, 2**36 + 5), (513, 2**40 + 15)):
            seed = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
            want, wa = orc.chacha12_fr_rand(curve, seed, pos, n)
            buf, ga = ctx.chacha12_fr_rand(curve, seed, pos, n)
            got = buf.download((n, 4)); buf.free()
            np.testing.assert_array_equal(got, want); assert ga == wa, (n, pos)
        # nothing to draw: the position stays
        buf, ga = ctx.chacha12_fr_rand(curve, bytes(32), 40, 0); buf.free()
        assert ga == 40
    finally:
        ctx.close()


@pytest.mark.gpu
def test_device_draws_without_the_wait():
    """cg_chacha12_fr_rand_dev_begin / _finish: several draws in flight on one context, finished in another order; the values and positions
    are those of the waiting call (and of the oracle); tickets are refused once used, and a ninth draw in flight is refused"""
    ensure_built()
    ctx = cg.Context()
    rng = np.random.default_rng(77)
    try:
        cases = [(BN254, 70001, 64), (BLS12_381, 12345, 2**36 + 5), (BN254, 1, 0), (BN254, 1 << 16, 7)]
        seeds = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in cases]
        inflight = [ctx.chacha12_fr_rand_begin(c, s, pos, n) for (c, n, pos), s in zip(cases, seeds)]
        # a consumer enqueued behind the draw on the same stream sees the values: add the drawn vector to itself before anything is waited for
        dbl = ctx.alloc(cases[0][1] * 32)
        cg._chk(cg.load().cg_vec_add_dev(ctx.h, BN254, cg.C.c_void_p(dbl.ptr), cg.C.c_void_p(inflight[0][0].ptr), cg.C.c_void_p(inflight[0][0].ptr), cg.C.c_size_t(cases[0][1])))
        for i in (2, 0, 3, 1):
            (c, n, pos), s = cases[i], seeds[i]
            after = ctx.chacha12_fr_rand_finish(inflight[i][1])
            want, wa = orc.chacha12_fr_rand(c, s, pos, n)
            np.testing.assert_array_equal(inflight[i][0].download((n, 4)), want); assert after == wa, i
        want0, _ = orc.chacha12_fr_rand(BN254, seeds[0], cases[0][2], cases[0][1])
        np.testing.assert_array_equal(dbl.download((cases[0][1], 4)), orc.field_op(BN254, FR, "add", want0, want0))
        with pytest.raises(cg.BackendError):
            ctx.chacha12_fr_rand_finish(inflight[0][1])                      # already finished
        with pytest.raises(cg.BackendError):
            ctx.chacha12_fr_rand_finish(99)
        for b, _ in inflight: b.free()
        dbl.free()
        many = [ctx.chacha12_fr_rand_begin(BN254, seeds[0], 0, 100) for _ in range(8)]
        with pytest.raises(cg.BackendError):
            ctx.chacha12_fr_rand_begin(BN254, seeds[0], 0, 100)               # eight in flight is the limit
        for b, tk in many: ctx.chacha12_fr_rand_finish(tk); b.free()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_device_draws_at_full_size():
    """2^22 draws (one masking vector of the headline circuit) against the host library's single-thread draws"""
    import time
    ensure_built()
    ctx = cg.Context()
    try:
        seed = bytes(range(7, 39)); n = 1 << 22
        t0 = time.time(); want, wa = cg.chacha12_fr_rand_host(BN254, seed, 0, n); t_host = time.time() - t0
        buf, ga = ctx.chacha12_fr_rand(BN254, seed, 0, n); buf.free()          # warm-up (allocations)
        t0 = time.time(); buf, ga = ctx.chacha12_fr_rand(BN254, seed, 0, n); t_dev = time.time() - t0
        got = buf.download((n, 4)); buf.free()
        np.testing.assert_array_equal(got, want); assert ga == wa
        print(f"2^22 F::rand draws: host (one thread) {t_host * 1e3:.1f} ms, device {t_dev * 1e3:.2f} ms")
    finally:
        ctx.close()


def rep3_share(curve, vals, rng):
    a = orc.random_field(curve, FR, vals.shape[0], rng); b = orc.random_field(curve, FR, vals.shape[0], rng)
    c = orc.field_op(curve, FR, "sub", orc.field_op(curve, FR, "sub", vals, a), b)
    return [a, b, c], [c, a, b]


@pytest.mark.gpu
@pytest.mark.parametrize("curve,log_m", [(BN254, 15), (BLS12_381, 14)])
def test_parties_with_device_drawn_masks_give_the_oracle_proof(curve, log_m, tmp_path):
    """three parties through the callback ABI, each holding Rep3Rand as two ChaCha12 seeds (rngs.rs:30-35; party i's rng2 is party
    i-1's rng1, rep3.rs:343-349): masks drawn on the GPU (generators described) and on the host (not described) give the same proofs,
    and both equal the oracle's, whose streams are the oracle's own draws from the same seeds; the generators end at the same positions"""
    ensure_built()
    threads = min(32, os.cpu_count() or 8)
    zp, wp = str(tmp_path / "s.zkey"), str(tmp_path / "s.wtns")
    orc.make_synthetic(curve, log_m, 41, zp, wp, threads=threads)
    z = orc.ZKey(curve, zp); w = orc.read_wtns(curve, wp)
    rng = np.random.default_rng(19)
    wa, wb = rep3_share(curve, w[2:], rng)
    seeds = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(3)]
    streams = [orc.chacha12_fr_rand(curve, s, 0, 2 * z.domain_size + 4)[0] for s in seeds]
    want = z.prove_rep3(w[:2], wa, wb, streams, threads=threads)
    sessions = {"one": cg.ProvingSession(curve, zp, precompute=False)}
    if curve == BN254: sessions["three contexts as devices"] = cg.ProvingSession(curve, zp, precompute=False, devices=[0, 0, 0], shared_devices=True)   # host/multidev.hpp: drawn on the primary, rows peer-copied
    try:
        results = {}
        for on_device, ses in ((True, sessions["one"]), (False, sessions["one"])) + (((None, sessions["three contexts as devices"]),) if len(sessions) > 1 else ()):
            hub = cg.LoopbackHub()
            rands = [cg.ChaChaRand(curve, seeds[i], seeds[(i + 2) % 3]) for i in range(3)]
            out, errs = [None] * 3, [None] * 3

            def party(i):
                try: out[i], _ = cg.host_prove_rep3_party(ses, w[:2], wa[i], wb[i], hub.net(i), rands[i].table, rands[i].streams if on_device is not False else None)
                except Exception as e: errs[i] = e; hub.abort()
            th = [threading.Thread(target=party, args=(i,)) for i in range(3)]
            for t in th: t.start()
            for t in th: t.join(300)
            assert errs == [None, None, None], errs
            results[on_device] = (np.stack(out), [r.positions() for r in rands])
            for r in rands: r.close()
            hub.close()
        np.testing.assert_array_equal(results[True][0], want)
        np.testing.assert_array_equal(results[False][0], want)
        assert results[True][1] == results[False][1]
        assert all(p1 >= 8 * 2 * z.domain_size for p1, _ in results[True][1])
        if None in results:
            np.testing.assert_array_equal(results[None][0], want); assert results[None][1] == results[True][1]
    finally:
        for ses in sessions.values(): ses.close()


@pytest.mark.gpu
def test_a_failed_proof_leaves_the_generators_behind_the_masks_it_drew(tmp_path):
    """Device draws are not waited for, and a mask may already be on its way to a peer when the proof dies (here: the first chunk of the first
    mul_vec message never leaves party 1).  Whatever path the proof leaves by, the caller's ChaCha12 generators must stand behind the words
    those masks were made from — a later proof from the same Rep3Rand must never repeat a mask (rep3.rs:656-660: the mask hides the local product)."""
    from test_rep3_party_abi import socket_ring
    ensure_built()
    curve, log_m = BN254, 15
    zp, wp = str(tmp_path / "s.zkey"), str(tmp_path / "s.wtns")
    orc.make_synthetic(curve, log_m, 47, zp, wp, threads=min(32, os.cpu_count() or 8))
    z = orc.ZKey(curve, zp); w = orc.read_wtns(curve, wp)
    rng = np.random.default_rng(31)
    wa, wb = rep3_share(curve, w[2:], rng)
    seeds = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(3)]
    ses = cg.ProvingSession(curve, zp, precompute=False)
    try:
        ends = socket_ring()
        ends[1].fail_send_next_at = 0
        rands = [cg.ChaChaRand(curve, seeds[i], seeds[(i + 2) % 3]) for i in range(3)]
        errs = [None] * 3

        def party(i):
            try: cg.host_prove_rep3_party(ses, w[:2], wa[i], wb[i], ends[i].table, rands[i].table, rands[i].streams)
            except Exception as e: errs[i] = e
            finally: ends[i].close()
        th = [threading.Thread(target=party, args=(i,)) for i in range(3)]
        for t in th: t.start()
        for t in th: t.join(120)
        assert not any(t.is_alive() for t in th)
        assert errs[1] is not None and "send_next failed" in str(errs[1])
        # both masking vectors of the witness map (2 x domain_size draws per generator, >= 8 words each) were drawn before anything was sent
        for i, r in enumerate(rands):
            if errs[i] is None: continue
            p1, p2 = r.positions()
            assert p1 >= 8 * 2 * z.domain_size and p2 >= 8 * 2 * z.domain_size, (i, p1, p2)
        for r in rands: r.close()
    finally:
        ses.close()


@pytest.mark.gpu
@pytest.mark.parametrize("curve,log_m,n,t", [(BN254, 13, 3, 1), (BLS12_381, 12, 5, 2)])
def test_seeded_shamir_parties_give_the_oracle_proof(curve, log_m, n, t, tmp_path):
    """cgh_session_prove_shamir_party_seeded: each party's private generator is a ChaCha12 stream run by the library from the party's seed —
    the preprocess(amount) batch of amount * (1 + 3t) draws on the GPU, the rest on the host, one stream in the reference's draw order
    (shamir.rs:923-1010).  The oracle proves with each party's stream = the oracle's own draws from the same seed."""
    from test_rep3_party_abi import shamir_mesh
    ensure_built()
    threads = min(32, os.cpu_count() or 8)
    zp, wp = str(tmp_path / "s.zkey"), str(tmp_path / "s.wtns")
    orc.make_synthetic(curve, log_m, 43, zp, wp, threads=threads)
    z = orc.ZKey(curve, zp); w = orc.read_wtns(curve, wp)
    rng = np.random.default_rng(29)
    wits = orc.shamir_share(curve, w[2:], n, t, rng)
    amount = (2 * z.domain_size + 8) // (t + 1) + 1                       # one proof never refills (tests/test_shamir.py::full_amount)
    assert amount * (1 + 3 * t) >= 1 << 14                                # the batch is drawn on the device
    seeds = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n)]
    length = (1024 + amount) * (1 + 3 * t) + t * (2 * z.domain_size + 8) + 64
    streams = [orc.chacha12_fr_rand(curve, s, 0, length)[0] for s in seeds]
    want = orc.prove_shamir(z, n, t, w[:2], wits, streams, threads=threads, preprocess=amount)
    ses = cg.ProvingSession(curve, zp, precompute=False)
    ends = shamir_mesh(n, streams)
    out, errs = [None] * n, [None] * n

    def party(i):
        try: out[i], _ = cg.host_prove_shamir_party_seeded(ses, t, w[:2], wits[i], ends[i].net, seeds[i], preprocess=amount)
        except Exception as e: errs[i] = e
    try:
        th = [threading.Thread(target=party, args=(i,)) for i in range(n)]
        for x in th: x.start()
        for x in th: x.join(300)
        assert errs == [None] * n, errs
        np.testing.assert_array_equal(np.stack(out), want)
        assert all(e.k == 0 for e in ends)                                 # the callback randomness was never asked
    finally:
        for e in ends: e.close()
        ses.close()


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_plonk_parties_with_device_drawn_masks(curve_name, host_option):
    """cgh_plonk_prove_rep3_party_ex: the mul_vec masks of rounds 2 and 3 (co-plonk/src/round2.rs, round3.rs) drawn on the GPU from the
    described generators (threshold lowered so that the fixture's vectors qualify) — the same proof as with host-drawn masks, the
    generators at the same positions afterwards, and the proof verifies; blinding drawn with rand() (round1.rs:93-99) in both runs"""
    from test_plonk_rounds import fx as pfx, CURVES, rep3_share as share3
    ensure_built()
    host_option(cg.HOST_OPT_DEVICE_MASKS_MIN, 64)
    curve = CURVES[curve_name]
    zp = pfx(curve_name, "circuit.zkey")
    npub = orc.plonk_zkey_info(curve, zp)["n_public"]
    w = orc.read_wtns(curve, pfx(curve_name, "witness.wtns"))
    rng = np.random.default_rng(77)
    wa, wb = share3(curve, w[npub + 1:], rng)
    seeds = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(3)]
    runs = {}
    for on_device in (True, False):
        hub = cg.LoopbackHub()
        rnds = [cg.ChaChaRand(curve, seeds[i], seeds[(i + 2) % 3]) for i in range(3)]
        got, errs = [None] * 3, [None] * 3

        def run(i):
            try: got[i] = cg.plonk_prove_rep3_party(curve, zp, w[:npub + 1], wa[i], wb[i], hub.net(i), rnds[i].table, None, None, upto=5,
                                                    streams_table=rnds[i].streams if on_device else None)
            except Exception as e: errs[i] = e; hub.abort()
        th = [threading.Thread(target=run, args=(i,)) for i in range(3)]
        for t in th: t.start()
        for t in th: t.join(300)
        assert errs == [None] * 3, errs
        runs[on_device] = (got, [r.positions() for r in rnds])
        for r in rnds: r.close()
        hub.close()
    for key in runs[True][0][0]:
        for party in range(3):
            np.testing.assert_array_equal(runs[True][0][party][key], runs[False][0][0][key], err_msg=f"{key} party {party}")
    assert runs[True][1] == runs[False][1]
    assert orc.plonk_verify(curve, zp, runs[True][0][0], w[1:npub + 1])


@pytest.mark.gpu
def test_seeded_shamir_parties_over_the_library_mesh(host_option):
    """the same entry over cgh_shamir_loopback_* (no Python in the data path) on the poseidon fixture, the device-draw threshold lowered so
    that both the preprocess batch and the king's re-sharing coefficients (shamir.rs:347-360: t draws per element, element by element) come
    from the kernels — the oracle's proof on the oracle's own draws from the same seeds"""
    from test_shamir import fx as sfx
    ensure_built()
    host_option(cg.HOST_OPT_DEVICE_MASKS_MIN, 32)
    curve, n, t = BN254, 5, 2
    zp = sfx("bn254", "poseidon", "circuit.zkey")
    z = orc.ZKey(curve, zp); w = orc.read_wtns(curve, sfx("bn254", "poseidon", "witness.wtns"))
    rng = np.random.default_rng(31)
    pub = w[:z.n_public + 1]
    wits = orc.shamir_share(curve, w[z.n_public + 1:], n, t, rng)
    amount = (2 * z.domain_size + 8) // (t + 1) + 1
    seeds = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n)]
    length = (1024 + amount) * (1 + 3 * t) + t * (2 * z.domain_size + 8) + 64
    streams = [orc.chacha12_fr_rand(curve, s, 0, length)[0] for s in seeds]
    want = orc.prove_shamir(z, n, t, pub, wits, streams, preprocess=amount)
    ses = cg.ProvingSession(curve, zp, precompute=False)
    hub = cg.ShamirLoopbackHub(n)
    nets = [hub.net(i) for i in range(n)]
    out, errs = [None] * n, [None] * n

    def party(i):
        try: out[i], _ = cg.host_prove_shamir_party_seeded(ses, t, pub, wits[i], nets[i], seeds[i], preprocess=amount)
        except Exception as e: errs[i] = e; hub.abort()
    try:
        th = [threading.Thread(target=party, args=(i,)) for i in range(n)]
        for x in th: x.start()
        for x in th: x.join(300)
        assert errs == [None] * n, errs
        np.testing.assert_array_equal(np.stack(out), want)
    finally:
        hub.close(); ses.close()


# ---- the day a Rust toolchain exists: rust/pin-vectors prints the reference crates' own values into tests/golden/rust_pins.json ------------
RUST_PINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rust_pins.json")
PINS_MISSING = ("tests/golden/rust_pins.json is absent: PARITY UNPINNED for the F::rand draw order (no Rust toolchain in the image that built this "
                "tree; `cargo run --release --manifest-path rust/pin-vectors/Cargo.toml -- tests/golden` next to a reference checkout writes it, rust/README.md)")


def _pins():
    import json
    if not os.path.exists(RUST_PINS):
        pytest.skip(PINS_MISSING)
    return json.load(open(RUST_PINS))


def _limbs(v):
    return np.array([int(x, 16) for x in v["montgomery_limbs_le"]], dtype=np.uint64)


@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_rust_pins_fr_rand_oracle_and_host(curve_name):
    """ark-ff's `Fr::rand(&mut ChaCha12Rng::from_seed(s))` as printed by the reference's crates: the oracle's and the host library's draws
    must be those values, from stream positions 0, unaligned and above 2^32 blocks, and end at the same word position"""
    pins = _pins()[curve_name]
    curve = BN254 if curve_name == "bn254" else BLS12_381
    ensure_built()
    for case in pins["fr_rand"]:
        seed, pos, want = bytes.fromhex(case["seed_hex"]), int(case["word_pos"]), np.stack([_limbs(v) for v in case["draws"]])
        got, after = orc.chacha12_fr_rand(curve, seed, pos, want.shape[0])
        np.testing.assert_array_equal(got, want); assert after == int(case["word_pos_after"])
        got, after = cg.chacha12_fr_rand_host(curve, seed, pos, want.shape[0])
        np.testing.assert_array_equal(got, want); assert after == int(case["word_pos_after"])
        for v, x in zip(case["draws"], got):                                   # the Montgomery limbs decode to the canonical integers printed beside them
            assert orc.to_dec(curve, FR, x) == v["canonical_decimal"]
    m = pins["rep3_masks"]
    s1, s2 = bytes.fromhex(m["seed1_hex"]), bytes.fromhex(m["seed2_hex"])
    want = np.stack([_limbs(v) for v in m["masking_field_elements"]])
    a, p1 = orc.chacha12_fr_rand(curve, s1, 0, want.shape[0]); b, p2 = orc.chacha12_fr_rand(curve, s2, 0, want.shape[0])
    np.testing.assert_array_equal(orc.field_op(curve, FR, "sub", a, b), want)          # Rep3Rand::masking_field_element, rngs.rs:37-40
    assert (p1, p2) == (int(m["word_pos1_after"]), int(m["word_pos2_after"]))


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_rust_pins_fr_rand_device(curve_name):
    """the same values from cg_chacha12_fr_rand_dev (the draws the product's timed region makes)"""
    pins = _pins()[curve_name]
    curve = BN254 if curve_name == "bn254" else BLS12_381
    ensure_built()
    ctx = cg.Context(0)
    try:
        for case in pins["fr_rand"]:
            seed, pos, want = bytes.fromhex(case["seed_hex"]), int(case["word_pos"]), np.stack([_limbs(v) for v in case["draws"]])
            buf, after = ctx.chacha12_fr_rand(curve, seed, pos, want.shape[0])
            np.testing.assert_array_equal(buf.download((want.shape[0], 4)), want); assert after == int(case["word_pos_after"])
            buf.free()
    finally:
        ctx.close()
