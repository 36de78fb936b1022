"""Pins the CPU oracle against every golden vector the reference's own tests hold for the Groth16 path:
zkey decode KATs, Fr product KAT, witness KAT, snarkjs proof KATs (verification), prove->verify on the four fixtures with
the plain driver and with three in-process REP3 parties (all three proofs equal)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as orc
from oracle_lib import BN254, BLS12_381, FR, FQ, G1, G2

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))
CURVES = {"bn254": BN254, "bls12_381": BLS12_381}
FIXTURES = [(c, k) for c in ("bn254", "bls12_381") for k in ("multiplier2", "poseidon")]


def fx(curve_name, circuit, f):
    return os.path.join(GOLDEN, "groth16", curve_name, circuit, f)


def pts_from_tokens(curve, group, toks):
    per = 2 if group == G1 else 4
    nq = orc.nlimbs(curve, FQ)
    out, i = [], 0
    while i < len(toks):
        if toks[i] == "inf":
            out.append(np.zeros(per * nq, dtype=np.uint64)); i += 1
        else:
            out.append(np.concatenate([orc.from_dec(curve, FQ, t) for t in toks[i:i + per]])); i += per
    return np.stack(out)


def test_field_constants_and_one_bytes():
    # zkey.rs:590-640: snarkjs dump of Fq.one / G1.one in Montgomery LE form
    b = KATS["bn254_one_bytes"]
    one = orc.from_dec(BN254, FQ, 1)
    assert one.tobytes() == bytes(b["fq_buf"])
    g1 = np.concatenate([orc.from_dec(BN254, FQ, 1), orc.from_dec(BN254, FQ, 2)])
    assert g1.tobytes() == bytes(b["g1_buf"])
    g2 = orc.generator_mul(BN254, G2, orc.from_dec(BN254, FR, 1))
    assert g2.tobytes() == bytes(b["g2_buf"])
    for curve in (BN254, BLS12_381):
        for which in (FR, FQ):
            p = orc.MODULI[(curve, which)]
            assert orc.to_dec(curve, which, orc.from_dec(curve, which, p - 1)) == str(p - 1)
            assert orc.to_dec(curve, which, orc.from_dec(curve, which, p + 5)) == "5"
        assert orc.on_curve(curve, G1, orc.generator_mul(curve, G1, orc.from_dec(curve, FR, 1)))
        assert orc.on_curve(curve, G2, orc.generator_mul(curve, G2, orc.from_dec(curve, FR, 1)))
        # generators have order r: (r-1)G + G = infinity
        for g in (G1, G2):
            m1 = orc.generator_mul(curve, g, orc.from_dec(curve, FR, orc.MODULI[(curve, FR)] - 1))
            assert not orc.point_add(curve, g, m1, orc.generator_mul(curve, g, orc.from_dec(curve, FR, 1))).any()


def test_fr_mul_kat_rep3_mul_vec_bn():
    # mpc-core/tests/protocols/rep3.rs:242-350: should_result[i] = x[i] * y[i]
    k = KATS["rep3_mul_vec_bn"]
    x = np.stack([orc.from_dec(BN254, FR, s) for s in k["x"]])
    y = np.stack([orc.from_dec(BN254, FR, s) for s in k["y"]])
    z = orc.field_op(BN254, FR, "mul", x, y)
    assert [orc.to_dec(BN254, FR, v) for v in z] == k["should_result"]
    # cross-check field ops against Python big ints
    rng = np.random.default_rng(1)
    for curve in (BN254, BLS12_381):
        for which in (FR, FQ):
            p = orc.MODULI[(curve, which)]; N = orc.nlimbs(curve, which)
            Rinv = pow(1 << (64 * N), -1, p)
            a = orc.random_field(curve, which, 50, rng); b = orc.random_field(curve, which, 50, rng)
            av = [orc.limbs_to_int(v) * Rinv % p for v in a]; bv = [orc.limbs_to_int(v) * Rinv % p for v in b]
            for op, f in (("add", lambda u, v: (u + v) % p), ("sub", lambda u, v: (u - v) % p), ("mul", lambda u, v: u * v % p)):
                got = orc.field_op(curve, which, op, a, b)
                assert [orc.limbs_to_int(v) * Rinv % p for v in got] == [f(u, v) for u, v in zip(av, bv)]
            assert orc.limbs_to_int(orc.field_inverse(curve, which, a[0])) * Rinv % p == pow(av[0], -1, p)


def test_snarkjs_roots_of_unity():
    # SURVEY.md §9 values (computed independently with Python big ints)
    q, roots, ta = orc.roots_of_unity(BN254)
    assert ta == 28 and orc.to_dec(BN254, FR, q) == "5"
    assert orc.to_dec(BN254, FR, roots[0]) == "1"
    assert orc.to_dec(BN254, FR, roots[1]) == str(orc.MODULI[(BN254, FR)] - 1)
    assert orc.to_dec(BN254, FR, roots[2]) == "21888242871839275217838484774961031246007050428528088939761107053157389710902"
    assert orc.to_dec(BN254, FR, roots[8]) == "3478517300119284901893091970156912948790432420133812234316178878452092729974"
    assert orc.to_dec(BN254, FR, roots[9]) == "6837567842312086091520287814181175430087169027974246751610506942214842701774"
    q, roots, ta = orc.roots_of_unity(BLS12_381)
    assert ta == 32 and orc.to_dec(BLS12_381, FR, q) == "5"
    for curve in (BN254, BLS12_381):
        p = orc.MODULI[(curve, FR)]
        q, roots, ta = orc.roots_of_unity(curve)
        t = (p - 1) >> ta
        z = pow(5, t, p)
        for i in range(ta + 1):
            assert orc.to_dec(curve, FR, roots[i]) == str(pow(z, 1 << (ta - i), p))


@pytest.mark.parametrize("name", ["can_deser_bn254_mult2_key", "can_deser_bls12_381_mult2_key"])
def test_zkey_decode_kat(name):
    k = KATS["zkey"][name]
    curve = BN254 if "bn254" in name else BLS12_381
    z = orc.ZKey(curve, os.path.join(GOLDEN, k["file"].replace("Groth16/", "groth16/")))
    v = k["values"]
    vk1 = z.points("vk_g1"); vk2 = z.points("vk_g2")
    np.testing.assert_array_equal(vk1[0], pts_from_tokens(curve, G1, v["alpha_g1"])[0])
    np.testing.assert_array_equal(vk1[1], pts_from_tokens(curve, G1, v["beta_g1"])[0])
    np.testing.assert_array_equal(vk1[2], pts_from_tokens(curve, G1, v["delta_g1"])[0])
    np.testing.assert_array_equal(vk2[0], pts_from_tokens(curve, G2, v["beta_g2"])[0])
    np.testing.assert_array_equal(vk2[1], pts_from_tokens(curve, G2, v["gamma_g2"])[0])
    np.testing.assert_array_equal(vk2[2], pts_from_tokens(curve, G2, v["delta_g2"])[0])
    for q, g in (("a_query", G1), ("b_g1_query", G1), ("b_g2_query", G2), ("h_query", G1), ("l_query", G1)):
        np.testing.assert_array_equal(z.points(q), pts_from_tokens(curve, g, v[q]), err_msg=q)
    np.testing.assert_array_equal(z.points("ic"), pts_from_tokens(curve, G1, v["gamma_abc_g1"]))
    assert (z.n_public + 1, z.n_vars - z.n_public, z.num_constraints, z.nnz_a, z.nnz_b) == (2, 3, 1, 1, 1)
    if "a" in v:   # zkey.rs:568-584 (bn254): A = [[(-1, 2)]], B = [[(1, 3)]]
        rp, col, co = z.matrix(0)
        assert list(rp) == [0, 1] and list(col) == [2] and orc.to_dec(curve, FR, co[0]) == v["a"][0]
        rp, col, co = z.matrix(1)
        assert list(rp) == [0, 1] and list(col) == [3] and orc.to_dec(curve, FR, co[0]) == v["b"][0]


@pytest.mark.parametrize("name", ["can_deser_witness_bn254", "can_deser_witness_bls12381"])
def test_witness_kat(name):
    k = KATS["witness"][name]
    curve = BN254 if "bn254" in name else BLS12_381
    w = orc.read_wtns(curve, os.path.join(GOLDEN, k["file"].replace("Groth16/", "groth16/")))
    assert [orc.to_dec(curve, FR, v) for v in w] == k["values"]


def test_ntt_matches_definition_and_roundtrip():
    rng = np.random.default_rng(7)
    for curve in (BN254, BLS12_381):
        _, roots, _ = orc.roots_of_unity(curve)
        for lg in (1, 2, 5, 8):
            n = 1 << lg
            x = orc.random_field(curve, FR, n, rng)
            f = orc.ntt(curve, x, roots[lg])
            np.testing.assert_array_equal(f, orc.dft_naive(curve, x, roots[lg]))
            np.testing.assert_array_equal(orc.ntt(curve, f, roots[lg], inverse=True), x)


def test_msm_pippenger_matches_naive():
    rng = np.random.default_rng(11)
    for curve in (BN254, BLS12_381):
        for group in (G1, G2):
            for n in (1, 5, 40):
                k = orc.random_field(curve, FR, n, rng)
                pts = np.stack([orc.generator_mul(curve, group, s) for s in k])
                sc = orc.random_field(curve, FR, n, rng)
                sc[0] = orc.from_dec(curve, FR, 1); 
                if n > 2:
                    sc[1] = 0; pts[2] = 0   # zero scalar, infinity base
                a = orc.msm(curve, group, pts, sc, "pippenger")
                b = orc.msm(curve, group, pts, sc, "naive")
                np.testing.assert_array_equal(a, b)
                # against the discrete logs: sum k_i s_i * G
                p = orc.MODULI[(curve, FR)]; Rinv = pow(1 << 256, -1, p)
                val = lambda v: orc.limbs_to_int(v) * Rinv % p
                e = sum(val(ki) * val(si) for i, (ki, si) in enumerate(zip(k, sc)) if not (n > 2 and i == 2)) % p
                np.testing.assert_array_equal(a, orc.generator_mul(curve, group, orc.from_dec(curve, FR, e)))


def test_pairing_bilinear():
    for curve in (BN254, BLS12_381):
        assert orc.pairing_selfcheck(curve, orc.from_dec(curve, FR, 123456789123456789))


@pytest.mark.parametrize("curve_name,circuit", FIXTURES)
def test_snarkjs_proof_kat_verifies(curve_name, circuit):
    """co-groth16/src/lib.rs:56-73,104-140 ; e2e_tests/mod.rs:85-100"""
    curve = CURVES[curve_name]
    vk = orc.vk_from_json(curve, fx(curve_name, circuit, "verification_key.json"))
    pub = orc.public_from_json(curve, fx(curve_name, circuit, "public.json"))
    proof = orc.proof_from_json(curve, fx(curve_name, circuit, "circom.proof"))
    assert orc.verify(curve, vk, pub, proof)
    bad = proof.copy(); bad[:orc.nlimbs(curve, FQ) * 2] = orc.generator_mul(curve, G1, orc.from_dec(curve, FR, 7))
    assert not orc.verify(curve, vk, pub, bad)
    if pub.shape[0]:
        pub2 = pub.copy(); pub2[0] = orc.from_dec(curve, FR, 34)
        assert not orc.verify(curve, vk, pub2, proof)
    # JSON round trip (proof.rs:8-29)
    assert orc.proof_to_json(curve, proof) == json.load(open(fx(curve_name, circuit, "circom.proof")))


def vk_of_zkey(z):
    v1 = z.points("vk_g1"); v2 = z.points("vk_g2")
    return {"alpha1": v1[0], "beta2": v2[0], "gamma2": v2[1], "delta2": v2[2], "ic": z.points("ic")}


@pytest.mark.parametrize("curve_name,circuit", FIXTURES)
def test_plain_prove_verifies(curve_name, circuit):
    """co-groth16/src/lib.rs:27-53,76-101,143-206"""
    curve = CURVES[curve_name]
    z = orc.ZKey(curve, fx(curve_name, circuit, "circuit.zkey"))
    w = orc.read_wtns(curve, fx(curve_name, circuit, "witness.wtns"))
    rng = np.random.default_rng(3)
    r, s = orc.random_field(curve, FR, 2, rng)
    proof = z.prove_plain(w, r, s)
    vk = orc.vk_from_json(curve, fx(curve_name, circuit, "verification_key.json"))
    zvk = vk_of_zkey(z)
    for k in ("alpha1", "beta2", "gamma2", "delta2", "ic"):
        np.testing.assert_array_equal(vk[k], zvk[k])
    pub = w[1:1 + z.n_public]
    np.testing.assert_array_equal(pub, orc.public_from_json(curve, fx(curve_name, circuit, "public.json")))
    assert orc.verify(curve, vk, pub, proof)
    # with r = s = 0 the proof is a deterministic function of (zkey, witness)
    zero = np.zeros(4, dtype=np.uint64)
    p0 = z.prove_plain(w, zero, zero)
    assert orc.verify(curve, vk, pub, p0)
    assert not np.array_equal(p0, proof)


def rep3_share(curve, vals, rng):
    """share_field_elements (rep3.rs:124-150)"""
    a = orc.random_field(curve, FR, vals.shape[0], rng); b = orc.random_field(curve, FR, vals.shape[0], rng)
    c = orc.field_op(curve, FR, "sub", orc.field_op(curve, FR, "sub", vals, a), b)
    return [a, b, c], [c, a, b]


@pytest.mark.parametrize("curve_name,circuit", FIXTURES)
def test_rep3_prove_three_parties_agree_and_verify(curve_name, circuit):
    """tests/tests/circom/e2e_tests/mod.rs:33-82"""
    curve = CURVES[curve_name]
    z = orc.ZKey(curve, fx(curve_name, circuit, "circuit.zkey"))
    w = orc.read_wtns(curve, fx(curve_name, circuit, "witness.wtns"))
    rng = np.random.default_rng(5)
    pub = w[:z.n_public + 1]
    wa, wb = rep3_share(curve, w[z.n_public + 1:], rng)
    streams = [orc.random_field(curve, FR, 2 * z.domain_size + 4, rng) for _ in range(3)]
    proofs, h = z.prove_rep3(pub, wa, wb, streams, want_h=True)
    np.testing.assert_array_equal(proofs[0], proofs[1]); np.testing.assert_array_equal(proofs[1], proofs[2])
    vk = orc.vk_from_json(curve, fx(curve_name, circuit, "verification_key.json"))
    assert orc.verify(curve, vk, w[1:1 + z.n_public], proofs[0])
    # equals the plain proof with r = sum r_i, s = sum s_i (r_i = stream_i[2m], s_i = stream_i[2m+1])
    m = z.domain_size
    add = lambda x, y: orc.field_op(curve, FR, "add", x, y)
    r = add(add(streams[0][2 * m], streams[1][2 * m]), streams[2][2 * m])
    s = add(add(streams[0][2 * m + 1], streams[1][2 * m + 1]), streams[2][2 * m + 1])
    np.testing.assert_array_equal(z.prove_plain(w, r, s), proofs[0])


@pytest.mark.parametrize("log_n", [1, 4, 14, 15, 16, 18])
def test_cpu_baseline_transforms_equal_the_plain_ones(log_n):
    """bench.py's cpu_baseline leg times cache-blocked, multi-threaded transforms (oracle/bench.hpp); they must produce what the
    plain radix-2 transforms of oracle/poly.hpp produce, below, at and above the block size"""
    for curve in (orc.BN254, orc.BLS12_381):
        assert orc.lib().orc_bench_ntt_selfcheck(curve, log_n, 4) == 1


@pytest.mark.parametrize("curve_name,circuit", [(c, k) for c in ("bn254", "bls12_381") for k in ("multiplier2", "poseidon")])
def test_vk_alphabeta_12_pairing_kat(curve_name, circuit):
    """`vk_alphabeta_12` of verification_key.json (circom-types/src/groth16/verification_key.rs:46-49) is e(vk_alpha_1, vk_beta_2), twelve
    base-field values written by snarkjs and parsed by the reference into `P::TargetField`: an exact-value KAT for the optimal ate
    pairing (Miller loop, Frobenius twist constants, final exponentiation with the libraries' exponent multiple, tower layout)."""
    curve = {"bn254": BN254, "bls12_381": BLS12_381}[curve_name]
    path = os.path.join(GOLDEN, "groth16", curve_name, circuit, "verification_key.json")
    vk = orc.vk_from_json(curve, path)
    want = json.load(open(path))["vk_alphabeta_12"]
    got = orc.pairing(curve, vk["alpha1"], vk["beta2"])
    for i in range(2):
        for j in range(3):
            for k in range(2):
                np.testing.assert_array_equal(got[i, j, k], orc.from_dec(curve, orc.FQ, want[i][j][k]), err_msg=f"c{i}.c{j}.c{k}")
    # and it is bilinear: e(P, Q) with P = alpha, Q = beta differs from e(2 alpha, beta) by a square (checked through e(alpha, 2 beta))
    two = orc.from_dec(curve, orc.FR, 2)
    a2 = orc.points_mul(curve, G1, vk["alpha1"][None], two[None])[0]
    b2 = orc.points_mul(curve, G2, vk["beta2"][None], two[None])[0]
    np.testing.assert_array_equal(orc.pairing(curve, a2, vk["beta2"]), orc.pairing(curve, vk["alpha1"], b2))
