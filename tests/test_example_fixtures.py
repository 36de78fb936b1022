"""Every Groth16 fixture the reference ships beyond test_vectors/Groth16: the example directory
(`co-circom/co-circom/examples/groth16/test_vectors/{kyc/bn254, kyc/bls12, poseidon, sum_arrays, multiplier2}`) and the criterion bench's key
(`test_vectors/benches/poseidon_hash2/bn254/groth16/poseidon.zkey`, tests/benches/poseidon_hash2.rs:175-223) — circuits with several public
inputs (kyc: 4, sum_arrays: 6), a circuit WITHOUT constraints and without private wires (sum_arrays), an output + a public input
(multiplier2), other matrix shapes.  zkey and verification key as shipped; witnesses as shipped where one ships, else derived from the
shipped .r1cs + input.json by tests/golden/make_example_witnesses.py (ORIGIN.json in each directory says which).
CPU: host readers == oracle readers; the oracle's plain and REP3 proofs verify under the SHIPPED verification keys (pins the witnesses).
GPU (-m gpu): plain + 3 x REP3 proofs bit-identical to the oracle's and verifying; the party entry on a validated session."""
import json
import os
import threading

import numpy as np
import pytest

import oracle_lib as orc
from oracle_lib import BN254, BLS12_381, FR
from product import cg, ensure_built

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CURVES = {"bn254": BN254, "bls12_381": BLS12_381}
EXAMPLES = [("bn254", "kyc"), ("bls12_381", "kyc"), ("bn254", "poseidon_hash2"), ("bn254", "sum_arrays"), ("bn254", "multiplier2_example")]
SHAPES = {("bn254", "kyc"): (17, 4, 16, 11), ("bls12_381", "kyc"): (17, 4, 16, 11), ("bn254", "poseidon_hash2"): (243, 1, 256, 240),
          ("bn254", "sum_arrays"): (7, 6, 8, 0), ("bn254", "multiplier2_example"): (4, 2, 4, 1)}      # n_vars, n_public, domain, constraints


def fx(curve_name, circuit, f):
    return os.path.join(GOLDEN, "groth16", curve_name, circuit, f)


def rep3_share(curve, vals, rng):
    a = orc.random_field(curve, FR, vals.shape[0], rng); b = orc.random_field(curve, FR, vals.shape[0], rng)
    c = orc.field_op(curve, FR, "sub", orc.field_op(curve, FR, "sub", vals, a), b)
    return [a, b, c], [c, a, b]


def load(curve_name, circuit, seed):
    curve = CURVES[curve_name]
    z = orc.ZKey(curve, fx(curve_name, circuit, "circuit.zkey")); w = orc.read_wtns(curve, fx(curve_name, circuit, "witness.wtns"))
    rng = np.random.default_rng(seed)
    pub = w[:z.n_public + 1]
    wa, wb = rep3_share(curve, w[z.n_public + 1:], rng)
    streams = [orc.random_field(curve, FR, 2 * z.domain_size + 4, rng) for _ in range(3)]
    vk = orc.vk_from_json(curve, fx(curve_name, circuit, "verification_key.json"))
    return curve, z, w, pub, wa, wb, streams, vk, rng


@pytest.mark.parametrize("curve_name,circuit", EXAMPLES)
def test_host_readers_match_the_oracle_on_the_examples(curve_name, circuit):
    ensure_built()
    curve = CURVES[curve_name]
    z = orc.ZKey(curve, fx(curve_name, circuit, "circuit.zkey"))
    assert (z.n_vars, z.n_public, z.domain_size, z.num_constraints) == SHAPES[(curve_name, circuit)]
    info = cg.host_zkey_info(curve, fx(curve_name, circuit, "circuit.zkey"))
    assert (info["n_vars"], info["n_public"], info["domain_size"], info["pow"], info["num_constraints"], info["nnz_a"], info["nnz_b"]) == \
           (z.n_vars, z.n_public, z.domain_size, z.pow, z.num_constraints, z.nnz_a, z.nnz_b)
    np.testing.assert_array_equal(cg.host_read_wtns(curve, fx(curve_name, circuit, "witness.wtns")), orc.read_wtns(curve, fx(curve_name, circuit, "witness.wtns")))
    with open(fx(curve_name, circuit, "ORIGIN.json")) as f:
        assert set(json.load(f)) == {"zkey", "witness"}


@pytest.mark.parametrize("curve_name,circuit", EXAMPLES)
def test_oracle_proofs_on_the_examples_verify_under_the_shipped_keys(curve_name, circuit):
    """pins the derived witnesses and the oracle on these shapes: Groth16::verify (co-groth16/src/groth16.rs:328-351) with the reference's own vk"""
    curve, z, w, pub, wa, wb, streams, vk, rng = load(curve_name, circuit, 21)
    r, s = orc.random_field(curve, FR, 2, rng)
    assert orc.verify(curve, vk, w[1:1 + z.n_public], z.prove_plain(w, r, s))
    proofs = z.prove_rep3(pub, wa, wb, streams)
    np.testing.assert_array_equal(proofs[0], proofs[1]); np.testing.assert_array_equal(proofs[1], proofs[2])
    assert orc.verify(curve, vk, w[1:1 + z.n_public], proofs[0])
    wrong = w[1:1 + z.n_public].copy(); wrong[0] = orc.field_op(curve, FR, "add", wrong[0:1], orc.from_dec(curve, FR, "1")[None])[0]
    assert not orc.verify(curve, vk, wrong, proofs[0])


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,circuit", EXAMPLES)
def test_plain_proof_on_the_examples(curve_name, circuit):
    ensure_built()
    curve, z, w, pub, wa, wb, streams, vk, rng = load(curve_name, circuit, 23)
    r, s = orc.random_field(curve, FR, 2, rng)
    proof, h = cg.prove_plain(curve, fx(curve_name, circuit, "circuit.zkey"), w, r, s, want_h=True)
    np.testing.assert_array_equal(h, z.witness_map_plain(w))
    np.testing.assert_array_equal(proof, z.prove_plain(w, r, s))
    assert orc.verify(curve, vk, w[1:1 + z.n_public], proof)


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,circuit", EXAMPLES)
def test_rep3_three_parties_on_the_examples(curve_name, circuit):
    """tests/tests/circom/e2e_tests/mod.rs:33-100 on the example circuits"""
    ensure_built()
    curve, z, w, pub, wa, wb, streams, vk, rng = load(curve_name, circuit, 25)
    proofs, h = cg.prove_rep3(curve, fx(curve_name, circuit, "circuit.zkey"), pub, wa, wb, streams, want_h=True)
    want, want_h = z.prove_rep3(pub, wa, wb, streams, want_h=True)
    np.testing.assert_array_equal(h, want_h)
    np.testing.assert_array_equal(proofs, want)
    assert orc.verify(curve, vk, w[1:1 + z.n_public], proofs[0])


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,circuit", EXAMPLES)
def test_party_entry_on_a_validated_session_of_the_examples(curve_name, circuit):
    """what `co-circom generate-proof` runs per process (co-circom.rs:484-506): the zkey goes through the session's GPU validation (on-curve +
    subgroup of every query point) and its window tables; three parties through the callback ABI give the oracle's proofs; the session's
    plain entry gives the oracle's plain proof"""
    ensure_built()
    curve, z, w, pub, wa, wb, streams, vk, rng = load(curve_name, circuit, 27)
    want = z.prove_rep3(pub, wa, wb, streams)
    ses = cg.ProvingSession(curve, fx(curve_name, circuit, "circuit.zkey"), precompute=True, validate=True)
    hub = cg.LoopbackHub()
    rands = [cg.StreamRand(curve, streams[i], streams[(i + 2) % 3]) for i in range(3)]
    try:
        out, errs = [None] * 3, [None] * 3

        def party(i):
            try: out[i], _ = cg.host_prove_rep3_party(ses, pub, wa[i], wb[i], hub.net(i), rands[i].table)
            except Exception as e: errs[i] = e
        th = [threading.Thread(target=party, args=(i,)) for i in range(3)]
        for t in th: t.start()
        for t in th: t.join(300)
        assert errs == [None, None, None], errs
        np.testing.assert_array_equal(np.stack(out), want)
        assert orc.verify(curve, vk, w[1:1 + z.n_public], out[0])
        r, s = orc.random_field(curve, FR, 2, rng)
        proof, _ = ses.prove_plain(w, r, s)
        np.testing.assert_array_equal(proof, z.prove_plain(w, r, s))
    finally:
        for x in rands: x.close()
        hub.close(); ses.close()


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,circuit", EXAMPLES)
def test_shamir_parties_on_the_examples(curve_name, circuit):
    """co-circom's second protocol (examples/groth16/run_full_kyc_shamir_bls.sh, run_full_poseidon_shamir.sh): three Shamir parties, threshold 1, on the
    example circuits == the oracle's Shamir proofs, verifying under the shipped keys"""
    ensure_built()
    curve = CURVES[curve_name]
    z = orc.ZKey(curve, fx(curve_name, circuit, "circuit.zkey")); w = orc.read_wtns(curve, fx(curve_name, circuit, "witness.wtns"))
    rng = np.random.default_rng(33)
    n, t = 3, 1
    wits = orc.shamir_share(curve, w[z.n_public + 1:], n, t, rng)
    need = (2 * z.domain_size + 4) // (1024 * (t + 1)) + 1
    streams = [orc.random_field(curve, FR, need * 1024 * (1 + 3 * t) + t * (2 * z.domain_size + 8), rng) for _ in range(n)]
    want = orc.prove_shamir(z, n, t, w[:z.n_public + 1], wits, streams)
    got = cg.prove_shamir(curve, fx(curve_name, circuit, "circuit.zkey"), n, t, w[:z.n_public + 1], wits, streams)
    np.testing.assert_array_equal(got, want)
    vk = orc.vk_from_json(curve, fx(curve_name, circuit, "verification_key.json"))
    assert orc.verify(curve, vk, w[1:1 + z.n_public], want[0])
