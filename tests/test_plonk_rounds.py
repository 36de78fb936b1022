"""co-plonk round 1 (first slice): wire-polynomial commitments = iNTT + MSM over p_tau.  The reference pins the EXACT
output for the deterministic blinding b_i = i (co-plonk/src/round1.rs:346-383) — the strongest parity evidence it has for the two
hot operations; the oracle is pinned to it on the CPU, the HIP path against both on the GPU."""
import json
import os

import numpy as np
import pytest

import oracle_lib as orc
from oracle_lib import BN254, BLS12_381, FR, FQ, G1
from product import cg, ensure_built

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL_KATS = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))
KATS = ALL_KATS["plonk_round1"]
CURVES = {"bn254": BN254, "bls12_381": BLS12_381}


def fx(curve_name, f):
    return os.path.join(GOLDEN, "plonk", curve_name, "multiplier2", f)


def deterministic_blinding(curve, count=6):
    return np.stack([orc.from_dec(curve, FR, str(i)) for i in range(count)])    # Round1Challenges::deterministic (round1.rs:99-107)


def kat_points(curve, kat):
    return np.stack([np.concatenate([orc.from_dec(curve, FQ, kat[k][0]), orc.from_dec(curve, FQ, kat[k][1])]) for k in ("commit_a", "commit_b", "commit_c")])


def test_oracle_round1_matches_reference_kat():
    kat = KATS["test_round1_multiplier2"]
    assert kat["file"] == "Plonk/bn254/multiplier2/circuit.zkey"
    info = orc.plonk_zkey_info(BN254, fx("bn254", "circuit.zkey"))
    assert (info["n_vars"], info["n_public"], info["n_constraints"]) == (4, 1, 2) or info["domain_size"] >= info["n_constraints"]
    w = orc.read_wtns(BN254, fx("bn254", "witness.wtns"))
    got = orc.plonk_round1_plain(BN254, fx("bn254", "circuit.zkey"), w, deterministic_blinding(BN254))
    np.testing.assert_array_equal(got, kat_points(BN254, kat))


@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_oracle_round1_structure(curve_name):
    """blinding only adds (b_hi X + b_lo)(X^n - 1): the commitment moves by b_lo (tau^n - 1) G + b_hi (tau^{n+1} - tau) G"""
    curve = CURVES[curve_name]
    zp = fx(curve_name, "circuit.zkey")
    info = orc.plonk_zkey_info(curve, zp)
    n = info["domain_size"]
    w = orc.read_wtns(curve, fx(curve_name, "witness.wtns"))
    _, _, _, p_tau = orc.plonk_zkey_data(curve, zp)
    for pt in p_tau:
        assert orc.on_curve(curve, G1, pt)
    zero = np.zeros((6, 4), dtype=np.uint64)
    c0, polys0 = orc.plonk_round1_plain(curve, zp, w, zero, want_polys=True)
    assert not polys0[:, n:, :].any()                                            # unblinded: degree < n
    b = deterministic_blinding(curve)
    c1, polys1 = orc.plonk_round1_plain(curve, zp, w, b, want_polys=True)
    for k in range(3):
        np.testing.assert_array_equal(polys1[k, n], b[2 * k + 1]); np.testing.assert_array_equal(polys1[k, n + 1], b[2 * k])
        # commitment recomputed from the coefficient vector with an independent (naive) MSM
        np.testing.assert_array_equal(orc.msm(curve, G1, p_tau[:n + 2], polys1[k], algo="naive"), c1[k])


def test_oracle_transcript_matches_reference_kat():
    """Keccak256 transcript (co-plonk/src/types.rs:194-227): points, the point at infinity and scalars -> challenge"""
    kat = ALL_KATS["plonk_transcript"]
    items = []
    for it in kat["items"]:
        if it[0] == "scalar": items.append(("scalar", orc.from_dec(BN254, FR, it[1])))
        elif it[0] == "point": items.append(("point", np.concatenate([orc.from_dec(BN254, FQ, it[1]), orc.from_dec(BN254, FQ, it[2])])))
        else: items.append(("point", np.zeros(8, dtype=np.uint64)))
    np.testing.assert_array_equal(orc.plonk_transcript(BN254, items), orc.from_dec(BN254, FR, kat["challenge"]))


def _transcript_items(curve):
    kat = ALL_KATS["plonk_transcript"]
    items = []
    for it in kat["items"]:
        if it[0] == "scalar": items.append(("scalar", orc.from_dec(curve, FR, it[1])))
        elif it[0] == "point": items.append(("point", np.concatenate([orc.from_dec(curve, FQ, it[1]), orc.from_dec(curve, FQ, it[2])])))
        else: items.append(("point", np.zeros(8, dtype=np.uint64)))
    return items, orc.from_dec(curve, FR, kat["challenge"])


def test_host_transcript_matches_reference_kat():
    """the host mirror's own Keccak256 transcript (no device needed) against the reference's KAT and, on random input, the oracle's"""
    ensure_built()
    items, want = _transcript_items(BN254)
    np.testing.assert_array_equal(cg.host_plonk_transcript(BN254, items), want)
    rng = np.random.default_rng(5)
    for curve in (BN254, BLS12_381):
        nq = 4 if curve == BN254 else 6
        more = [("scalar", s) for s in orc.random_field(curve, FR, 40, rng)] + [("point", np.zeros(2 * nq, dtype=np.uint64))]
        more += [("point", orc.generator_mul(curve, G1, s)) for s in orc.random_field(curve, FR, 3, rng)]
        np.testing.assert_array_equal(cg.host_plonk_transcript(curve, more), orc.plonk_transcript(curve, more))


def test_oracle_round2_matches_reference_kat():
    """[z]_1 of the grand-product polynomial with the deterministic blinding (co-plonk/src/round2.rs:326-355)"""
    kat = ALL_KATS["plonk_round2"]["test_round2_multiplier2"]
    w = orc.read_wtns(BN254, fx("bn254", "witness.wtns"))
    beta, gamma, cz = orc.plonk_round2_plain(BN254, fx("bn254", "circuit.zkey"), w, deterministic_blinding(BN254, 9))
    np.testing.assert_array_equal(cz, np.concatenate([orc.from_dec(BN254, FQ, kat["commit_z"][0]), orc.from_dec(BN254, FQ, kat["commit_z"][1])]))
    assert beta.any() and gamma.any() and not np.array_equal(beta, gamma)


def _pt(curve, xy):
    return np.concatenate([orc.from_dec(curve, FQ, xy[0]), orc.from_dec(curve, FQ, xy[1])])


def check_against_reference_kats(r):
    """every value the reference hard-codes for the five rounds on Plonk/bn254/multiplier2 with the deterministic blinding b_i = i
    (round1.rs:346-383, round2.rs:326-355, round3.rs:553-596, round4.rs:169-246, round5.rs:391-429)"""
    k1, k2, k3 = KATS["test_round1_multiplier2"], ALL_KATS["plonk_round2"]["test_round2_multiplier2"], ALL_KATS["plonk_round3"]["test_round3_multiplier2"]
    k4, k5 = ALL_KATS["plonk_round4"]["test_round4_multiplier2"], ALL_KATS["plonk_round5"]["test_round5_multiplier2"]
    for name, kat in (("a", k1["commit_a"]), ("b", k1["commit_b"]), ("c", k1["commit_c"]), ("z", k2["commit_z"]),
                      ("t1", k3["commit_t1"]), ("t2", k3["commit_t2"]), ("t3", k3["commit_t3"]), ("wxi", k5["wxi"]), ("wxiw", k5["wxiw"])):
        np.testing.assert_array_equal(r[name], _pt(BN254, kat), err_msg=name)
    for name in ("eval_a", "eval_b", "eval_c", "eval_zw", "eval_s1", "eval_s2"):
        np.testing.assert_array_equal(r[name], orc.from_dec(BN254, FR, k4[name]), err_msg=name)


def test_oracle_all_rounds_match_reference_kats():
    """the round-by-round prover state machine, ONE run through rounds 1-5"""
    w = orc.read_wtns(BN254, fx("bn254", "witness.wtns"))
    check_against_reference_kats(orc.plonk_prove_plain(BN254, fx("bn254", "circuit.zkey"), w, deterministic_blinding(BN254, 11), upto=5))


@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_oracle_verifier_accepts_snarkjs_proofs(curve_name):
    """pins the verifier (challenge derivation, linearisation, KZG pairing check) to proofs snarkjs produced: the reference verifies the
    same files (co-plonk/src/plonk.rs:352-366); tampering with any element must be rejected; the zkey's verifying key equals
    verification_key.json"""
    curve = CURVES[curve_name]
    zp = fx(curve_name, "circuit.zkey")
    proof = orc.plonk_proof_from_json(curve, fx(curve_name, "circom.proof"))
    pub = orc.public_from_json(curve, fx(curve_name, "public.json"))
    assert orc.plonk_verify(curve, zp, proof, pub)
    vkj = json.load(open(fx(curve_name, "verification_key.json")))
    vk = orc.plonk_zkey_vk(curve, zp)
    for key in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
        np.testing.assert_array_equal(vk[key], orc.g1_from_json(curve, vkj[key]), err_msg=key)
    np.testing.assert_array_equal(vk["X_2"], orc.g2_from_json(curve, vkj["X_2"]))
    np.testing.assert_array_equal(vk["k1"], orc.from_dec(curve, FR, vkj["k1"])); np.testing.assert_array_equal(vk["k2"], orc.from_dec(curve, FR, vkj["k2"]))
    for key in ("eval_a", "eval_zw"):
        bad = dict(proof); bad[key] = orc.field_op(curve, FR, "add", proof[key][None, :], orc.from_dec(curve, FR, "1")[None, :])[0]
        assert not orc.plonk_verify(curve, zp, bad, pub)
    bad = dict(proof); bad["wxi"] = proof["wxiw"]
    assert not orc.plonk_verify(curve, zp, bad, pub)
    assert not orc.plonk_verify(curve, zp, proof, pub[::-1].copy())


@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_oracle_prover_output_verifies(curve_name):
    curve = CURVES[curve_name]
    zp = fx(curve_name, "circuit.zkey")
    w = orc.read_wtns(curve, fx(curve_name, "witness.wtns"))
    npub = orc.plonk_zkey_info(curve, zp)["n_public"]
    proof = orc.plonk_prove_plain(curve, zp, w, orc.random_field(curve, FR, 11, np.random.default_rng(8)), upto=5)
    assert orc.plonk_verify(curve, zp, proof, w[1:npub + 1])


@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_host_plonk_proof_json_codec(curve_name):
    """host mirror's PlonkProof JSON codec against the snarkjs files the reference deserialises (circom-types/src/plonk/proof.rs:84-140)"""
    ensure_built()
    curve = CURVES[curve_name]
    text = open(fx(curve_name, "circom.proof")).read()
    got = cg.host_plonk_proof_from_json(curve, text)
    want = orc.plonk_proof_from_json(curve, fx(curve_name, "circom.proof"))
    for key in want:
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    assert json.loads(cg.host_plonk_proof_to_json(curve, got)) == json.loads(text)
    with pytest.raises(cg.BackendError):
        cg.host_plonk_proof_from_json(BLS12_381 if curve == BN254 else BN254, text)


def test_host_plonk_zkey_reader_matches_oracle():
    ensure_built()
    for name, curve in CURVES.items():
        assert cg.host_plonk_zkey_info(curve, fx(name, "circuit.zkey")) == orc.plonk_zkey_info(curve, fx(name, "circuit.zkey"))
    with pytest.raises(cg.BackendError):
        cg.host_plonk_zkey_info(BN254, os.path.join(GOLDEN, "groth16", "bn254", "multiplier2", "circuit.zkey"))    # a Groth16 zkey


def rep3_share(curve, vals, rng):
    a = orc.random_field(curve, FR, vals.shape[0], rng); b = orc.random_field(curve, FR, vals.shape[0], rng)
    c = orc.field_op(curve, FR, "sub", orc.field_op(curve, FR, "sub", vals, a), b)
    return [a, b, c], [c, a, b]                                  # party i holds (x_i, x_{i-1})


@pytest.mark.gpu
def test_gpu_round1_plain_matches_reference_kat():
    """the HIP path reproduces the reference's hard-coded commitments bit for bit (round1.rs:346-383)"""
    ensure_built()
    w = orc.read_wtns(BN254, fx("bn254", "witness.wtns"))
    got = cg.plonk_round1_plain(BN254, fx("bn254", "circuit.zkey"), w, deterministic_blinding(BN254))
    np.testing.assert_array_equal(got, kat_points(BN254, KATS["test_round1_multiplier2"]))


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_gpu_round1_plain_and_rep3_match_oracle(curve_name):
    ensure_built()
    curve = CURVES[curve_name]
    zp = fx(curve_name, "circuit.zkey")
    info = orc.plonk_zkey_info(curve, zp)
    w = orc.read_wtns(curve, fx(curve_name, "witness.wtns"))
    rng = np.random.default_rng(77)
    blind = orc.random_field(curve, FR, 6, rng)
    want = orc.plonk_round1_plain(curve, zp, w, blind)
    np.testing.assert_array_equal(cg.plonk_round1_plain(curve, zp, w, blind), want)
    # REP3: random sharings of the private witness and of the blinding values; every party must open the same three points
    npub = info["n_public"]
    wa, wb = rep3_share(curve, w[npub + 1:], rng)
    ba, bb = rep3_share(curve, blind, rng)
    got = cg.plonk_round1_rep3(curve, zp, w[:npub + 1], wa, wb, ba, bb)
    for party in range(3):
        np.testing.assert_array_equal(got[party], want, err_msg=f"party {party}")
    if curve == BN254:   # the reference's deterministic blinding as trivial shares (promote_to_trivial_share: ID0 -> a, ID1 -> b)
        det = deterministic_blinding(curve); zero = np.zeros_like(det)
        got = cg.plonk_round1_rep3(curve, zp, w[:npub + 1], wa, wb, [det, zero, zero], [zero, det, zero])
        for party in range(3):
            np.testing.assert_array_equal(got[party], kat_points(BN254, KATS["test_round1_multiplier2"]))


@pytest.mark.gpu
def test_gpu_round2_plain_matches_reference_kat():
    """[z]_1 from the HIP path (grand product as device vector kernels, iNTT, MSM) equals the reference's hard-coded point (round2.rs:326-355)"""
    ensure_built()
    kat = ALL_KATS["plonk_round2"]["test_round2_multiplier2"]
    w = orc.read_wtns(BN254, fx("bn254", "witness.wtns"))
    beta, gamma, cz = cg.plonk_round2_plain(BN254, fx("bn254", "circuit.zkey"), w, deterministic_blinding(BN254, 9))
    np.testing.assert_array_equal(cz, np.concatenate([orc.from_dec(BN254, FQ, kat["commit_z"][0]), orc.from_dec(BN254, FQ, kat["commit_z"][1])]))
    ob, og, _ = orc.plonk_round2_plain(BN254, fx("bn254", "circuit.zkey"), w, deterministic_blinding(BN254, 9))
    np.testing.assert_array_equal(beta, ob); np.testing.assert_array_equal(gamma, og)


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_gpu_round2_plain_matches_oracle(curve_name):
    ensure_built()
    curve = CURVES[curve_name]
    zp = fx(curve_name, "circuit.zkey")
    w = orc.read_wtns(curve, fx(curve_name, "witness.wtns"))
    blind = orc.random_field(curve, FR, 9, np.random.default_rng(123))
    want = orc.plonk_round2_plain(curve, zp, w, blind, want_poly=True)
    got = cg.plonk_round2_plain(curve, zp, w, blind, want_poly=True)
    for a, b, name in zip(got, want, ("beta", "gamma", "commit_z", "poly_z")):
        np.testing.assert_array_equal(a, b, err_msg=name)


@pytest.mark.gpu
def test_gpu_all_rounds_match_reference_kats():
    """one run of the HIP plain driver through rounds 1-5 reproduces every value the reference hard-codes: nine commitments, six evaluations"""
    ensure_built()
    w = orc.read_wtns(BN254, fx("bn254", "witness.wtns"))
    check_against_reference_kats(cg.plonk_prove_plain(BN254, fx("bn254", "circuit.zkey"), w, deterministic_blinding(BN254, 11), upto=5))


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_gpu_all_rounds_match_oracle(curve_name):
    """random blinding, both curves (BASELINE configs[4]: BLS12-381 co-plonk): the whole proof, the challenges and the quotient parts"""
    ensure_built()
    curve = CURVES[curve_name]
    zp = fx(curve_name, "circuit.zkey")
    w = orc.read_wtns(curve, fx(curve_name, "witness.wtns"))
    blind = orc.random_field(curve, FR, 11, np.random.default_rng(321))
    want = orc.plonk_prove_plain(curve, zp, w, blind, upto=5, want_t=True)
    got = cg.plonk_prove_plain(curve, zp, w, blind, upto=5, want_t=True)
    for key in want:
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    npub = orc.plonk_zkey_info(curve, zp)["n_public"]
    assert orc.plonk_verify(curve, zp, got, w[1:npub + 1])                            # the GPU's proof passes the (snarkjs-pinned) verifier


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_gpu_rep3_all_rounds_match_plain_oracle(curve_name):
    """three REP3 parties run the whole prover (mul_vec, array_prod_mul, inv_many, mul_open_many, open_many over the in-process
    network): every party must report the values the plain oracle computes from the witness and the OPENED blinding values — the
    masks and random shares of the protocols cancel.  On BN254 with the reference's deterministic blinding as trivial shares this
    is again the full set of hard-coded reference values."""
    ensure_built()
    curve = CURVES[curve_name]
    zp = fx(curve_name, "circuit.zkey")
    info = orc.plonk_zkey_info(curve, zp)
    npub = info["n_public"]
    w = orc.read_wtns(curve, fx(curve_name, "witness.wtns"))
    rng = np.random.default_rng(2024)
    blind = orc.random_field(curve, FR, 11, rng)
    wa, wb = rep3_share(curve, w[npub + 1:], rng)
    ba, bb = rep3_share(curve, blind, rng)
    streams = [orc.random_field(curve, FR, 40000, rng) for _ in range(3)]
    want = orc.plonk_prove_plain(curve, zp, w, blind, upto=5)
    got = cg.plonk_prove_rep3(curve, zp, w[:npub + 1], wa, wb, ba, bb, streams, upto=5)
    for party in range(3):
        for key in want:
            np.testing.assert_array_equal(got[party][key], want[key], err_msg=f"party {party} {key}")
    if curve == BN254:
        det = deterministic_blinding(curve, 11); zero = np.zeros_like(det)
        got = cg.plonk_prove_rep3(curve, zp, w[:npub + 1], wa, wb, [det, zero, zero], [zero, det, zero], streams, upto=5)
        for party in range(3):
            check_against_reference_kats(got[party])


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,n,t", [("bn254", 3, 1), ("bls12_381", 3, 1), ("bn254", 5, 2)])
def test_gpu_shamir_all_rounds_match_plain_oracle(curve_name, n, t):
    """n Shamir parties (threshold t): double-sharing generation, degree reduction after every product, array_prod_mul / inv_many /
    mul_open_many with 2t + 1 shares, openings "in circles" — every party must report the plain oracle's values for the opened blinding"""
    ensure_built()
    curve = CURVES[curve_name]
    zp = fx(curve_name, "circuit.zkey")
    npub = orc.plonk_zkey_info(curve, zp)["n_public"]
    w = orc.read_wtns(curve, fx(curve_name, "witness.wtns"))
    rng = np.random.default_rng(77 + n)
    blind = orc.random_field(curve, FR, 11, rng)
    wits = orc.shamir_share(curve, w[npub + 1:], n, t, rng)
    blinds = orc.shamir_share(curve, blind, n, t, rng)
    streams = [orc.random_field(curve, FR, 4 * 1024 * (1 + 3 * t) + 20000, rng) for _ in range(n)]
    want = orc.plonk_prove_plain(curve, zp, w, blind, upto=5)
    got = cg.plonk_prove_shamir(curve, zp, n, t, w[:npub + 1], wits, blinds, streams, upto=5)
    for party in range(n):
        for key in want:
            np.testing.assert_array_equal(got[party][key], want[key], err_msg=f"party {party} {key}")


# ---- ONE REP3 party of co-plonk behind the callback ABI (cgh_plonk_prove_rep3_party; co-circom.rs:560-600) -------------------------------
def test_plonk_party_entry_rejects_bad_arguments_without_a_gpu():
    """null tables / pointers, blinding shares given by halves, a round outside 1..5: status 1 and a message before any device is touched"""
    import ctypes as C
    ensure_built()
    h = cg.load_host()
    net = cg.Rep3NetTable(); rnd = cg.Rep3RandTable()
    out = np.zeros((9, 8), dtype=np.uint64); buf = np.zeros((16, 4), dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    zp = fx("bn254", "circuit.zkey").encode()
    assert h.cgh_plonk_prove_rep3_party(0, BN254, zp, p(buf), p(buf), p(buf), None, None, None, C.byref(rnd), 5, p(out), None, None) != 0
    assert b"null argument" in h.cgh_last_error()
    assert h.cgh_plonk_prove_rep3_party(0, BN254, zp, p(buf), p(buf), p(buf), p(buf), None, C.byref(net), C.byref(rnd), 5, p(out), None, None) != 0
    assert b"go together" in h.cgh_last_error()
    assert h.cgh_plonk_prove_rep3_party(0, BN254, zp, p(buf), p(buf), p(buf), None, None, C.byref(net), C.byref(rnd), 6, p(out), None, None) != 0
    assert b"upto" in h.cgh_last_error()


def _three_plonk_parties(curve, zp, pub, wa, wb, streams, blind=None, upto=5):
    """three threads, each ONE party through the callback ABI: loopback transport, randomness = the streams (party i: S_i, S_{i-1})"""
    import threading
    hub = cg.LoopbackHub()
    rnds = [cg.StreamRand(curve, streams[i], streams[(i + 2) % 3]) for i in range(3)]
    got, errs = [None] * 3, [None] * 3

    def run(i):
        try:
            ba, bb = (None, None) if blind is None else (blind[0][i], blind[1][i])
            got[i] = cg.plonk_prove_rep3_party(curve, zp, pub, wa[i], wb[i], hub.net(i), rnds[i].table, ba, bb, upto=upto)
        except Exception as e:                                                          # noqa: BLE001 (reported below, peers released)
            errs[i] = e; hub.abort()
    th = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    for t in th: t.start()
    for t in th: t.join()
    for r in rnds: r.close()
    hub.close()
    assert errs == [None] * 3, errs
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_gpu_plonk_party_entry_matches_plain_oracle(curve_name):
    """every party, alone behind the callback tables, reports the plain oracle's proof for the opened blinding — with the blinding
    shares handed in, and with the blinding drawn by rand() (round1.rs:93-99: eleven draws before anything else, so that
    b_t = S_0[t] + S_1[t] + S_2[t]); on BN254 the reference's deterministic blinding gives its hard-coded values again"""
    ensure_built()
    curve = CURVES[curve_name]
    zp = fx(curve_name, "circuit.zkey")
    npub = orc.plonk_zkey_info(curve, zp)["n_public"]
    w = orc.read_wtns(curve, fx(curve_name, "witness.wtns"))
    rng = np.random.default_rng(4242)
    blind = orc.random_field(curve, FR, 11, rng)
    wa, wb = rep3_share(curve, w[npub + 1:], rng)
    ba, bb = rep3_share(curve, blind, rng)
    streams = [orc.random_field(curve, FR, 40000, rng) for _ in range(3)]
    want = orc.plonk_prove_plain(curve, zp, w, blind, upto=5)
    got = _three_plonk_parties(curve, zp, w[:npub + 1], wa, wb, streams, (ba, bb))
    for party in range(3):
        for key in want:
            np.testing.assert_array_equal(got[party][key], want[key], err_msg=f"party {party} {key}")
    # the same values as the three-party in-process entry on the same streams
    ref3 = cg.plonk_prove_rep3(curve, zp, w[:npub + 1], wa, wb, ba, bb, streams, upto=5)
    for key in want:
        np.testing.assert_array_equal(got[0][key], ref3[0][key], err_msg=key)
    # blinding drawn through the randomness table
    drawn = orc.field_op(curve, FR, "add", orc.field_op(curve, FR, "add", streams[0][:11], streams[1][:11]), streams[2][:11])
    want = orc.plonk_prove_plain(curve, zp, w, drawn, upto=5)
    got = _three_plonk_parties(curve, zp, w[:npub + 1], wa, wb, streams, None)
    for party in range(3):
        for key in want:
            np.testing.assert_array_equal(got[party][key], want[key], err_msg=f"drawn blinding, party {party} {key}")
    assert orc.plonk_verify(curve, zp, got[0], w[1:npub + 1])
    if curve == BN254:
        det = deterministic_blinding(curve, 11); zero = np.zeros_like(det)
        got = _three_plonk_parties(curve, zp, w[:npub + 1], wa, wb, streams, ([det, zero, zero], [zero, det, zero]))
        for party in range(3):
            check_against_reference_kats(got[party])
