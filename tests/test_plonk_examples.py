"""co-plonk on the PLONK keys of the reference's example circuits — `co-circom/co-circom/examples/plonk/test_vectors/{kyc/bn254,
kyc/bls12, multiplier2, sum_arrays}` (data under tests/golden/plonk/, made by tests/golden/make_example_witnesses.py; the Poseidon example ships no
plonk zkey).  Beyond test_vectors/Plonk/*/multiplier2 these bring the zkey's ADDITIONS section (kyc: 19 additions — wires that are linear
combinations of others, co-plonk/src/types.rs), 4 and 6 public inputs and a circuit whose only gates are public-input gates (sum_arrays).
CPU: host zkey reader == oracle, the zkey's verifying key == the shipped verification_key.json, the oracle's proof passes the (snarkjs-pinned) oracle
verifier.  GPU (-m gpu): every value of the plain prover == the oracle's; three REP3 parties report the plain oracle's values; both verify."""
import json
import os

import numpy as np
import pytest

import oracle_lib as orc
from oracle_lib import BN254, BLS12_381, FR
from product import cg, ensure_built

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CURVES = {"bn254": BN254, "bls12_381": BLS12_381}
EXAMPLES = [("bn254", "kyc"), ("bls12_381", "kyc"), ("bn254", "sum_arrays"), ("bn254", "multiplier2_example")]
SHAPES = {("bn254", "kyc"): (36, 4, 64, 19, 34), ("bls12_381", "kyc"): (36, 4, 64, 19, 34), ("bn254", "sum_arrays"): (7, 6, 8, 0, 6),
          ("bn254", "multiplier2_example"): (4, 2, 8, 0, 3)}                    # n_vars, n_public, domain, additions, constraints


def fx(curve_name, circuit, f):
    return os.path.join(GOLDEN, "plonk", curve_name, circuit, f)


def rep3_share(curve, vals, rng):
    a = orc.random_field(curve, FR, vals.shape[0], rng); b = orc.random_field(curve, FR, vals.shape[0], rng)
    c = orc.field_op(curve, FR, "sub", orc.field_op(curve, FR, "sub", vals, a), b)
    return [a, b, c], [c, a, b]


@pytest.mark.parametrize("curve_name,circuit", EXAMPLES)
def test_oracle_and_host_reader_on_the_plonk_examples(curve_name, circuit):
    ensure_built()
    curve = CURVES[curve_name]
    zp = fx(curve_name, circuit, "circuit.zkey")
    info = orc.plonk_zkey_info(curve, zp)
    assert (info["n_vars"], info["n_public"], info["domain_size"], info["n_additions"], info["n_constraints"]) == SHAPES[(curve_name, circuit)]
    assert cg.host_plonk_zkey_info(curve, zp) == info
    vkj = json.load(open(fx(curve_name, circuit, "verification_key.json")))
    vk = orc.plonk_zkey_vk(curve, zp)
    for key in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
        np.testing.assert_array_equal(vk[key], orc.g1_from_json(curve, vkj[key]), err_msg=key)
    np.testing.assert_array_equal(vk["X_2"], orc.g2_from_json(curve, vkj["X_2"]))
    assert vkj["nPublic"] == info["n_public"]
    w = orc.read_wtns(curve, fx(curve_name, circuit, "witness.wtns"))
    proof = orc.plonk_prove_plain(curve, zp, w, orc.random_field(curve, FR, 11, np.random.default_rng(8)), upto=5)
    pub = w[1:info["n_public"] + 1]
    assert orc.plonk_verify(curve, zp, proof, pub)
    wrong = pub.copy(); wrong[0] = orc.field_op(curve, FR, "add", wrong[0:1], orc.from_dec(curve, FR, "1")[None])[0]
    assert not orc.plonk_verify(curve, zp, proof, wrong)


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,circuit", EXAMPLES)
def test_gpu_plain_and_rep3_on_the_plonk_examples(curve_name, circuit):
    ensure_built()
    curve = CURVES[curve_name]
    zp = fx(curve_name, circuit, "circuit.zkey")
    npub = orc.plonk_zkey_info(curve, zp)["n_public"]
    w = orc.read_wtns(curve, fx(curve_name, circuit, "witness.wtns"))
    rng = np.random.default_rng(77)
    blind = orc.random_field(curve, FR, 11, rng)
    want = orc.plonk_prove_plain(curve, zp, w, blind, upto=5, want_t=True)
    got = cg.plonk_prove_plain(curve, zp, w, blind, upto=5, want_t=True)
    for key in want:
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    assert orc.plonk_verify(curve, zp, got, w[1:npub + 1])
    wa, wb = rep3_share(curve, w[npub + 1:], rng)
    ba, bb = rep3_share(curve, blind, rng)
    streams = [orc.random_field(curve, FR, 40000, rng) for _ in range(3)]
    want = orc.plonk_prove_plain(curve, zp, w, blind, upto=5)
    parties = cg.plonk_prove_rep3(curve, zp, w[:npub + 1], wa, wb, ba, bb, streams, upto=5)
    for party in range(3):
        for key in want:
            np.testing.assert_array_equal(parties[party][key], want[key], err_msg=f"party {party} {key}")
