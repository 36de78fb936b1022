"""Shamir-shared co-groth16 (mpc-core/src/protocols/shamir.rs): oracle simulation on the CPU; ShamirHipProtocol parties on the GPU
must reproduce it bit for bit on the same randomness streams.  The reference snapshot has no Shamir Groth16 test to pin values
against, so the anchors are: proofs verify (pairing), all parties agree, and HIP == oracle."""
import os

import numpy as np
import pytest

import oracle_lib as orc
from oracle_lib import BN254, BLS12_381, FR
from product import cg, ensure_built

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CURVES = {"bn254": BN254, "bls12_381": BLS12_381}
FIXTURES = [(c, k) for c in ("bn254", "bls12_381") for k in ("multiplier2", "poseidon")]


def fx(curve_name, circuit, f):
    return os.path.join(GOLDEN, "groth16", curve_name, circuit, f)


def vk_of(z):
    v1 = z.points("vk_g1"); v2 = z.points("vk_g2")
    return {"alpha1": v1[0], "beta2": v2[0], "gamma2": v2[1], "delta2": v2[2], "ic": z.points("ic")}


def setup(curve_name, circuit, n, t, seed, preprocess=0):
    curve = CURVES[curve_name]
    z = orc.ZKey(curve, fx(curve_name, circuit, "circuit.zkey")); w = orc.read_wtns(curve, fx(curve_name, circuit, "witness.wtns"))
    rng = np.random.default_rng(seed)
    wits = orc.shamir_share(curve, w[z.n_public + 1:], n, t, rng)
    # per party: one batch of 1024 double sharings costs 1024 * (1 + 3t) draws; the king adds t per re-shared element
    need = (2 * z.domain_size + 4) // (1024 * (t + 1)) + 1
    streams = [orc.random_field(curve, FR, (need * 1024 + preprocess) * (1 + 3 * t) + t * (2 * z.domain_size + 8), rng) for _ in range(n)]
    return curve, z, w, wits, streams


def full_amount(z, t):
    """secrets to double-share so that one proof never refills: two mul_vec over the domain + the O(1) scalar pairs"""
    return (2 * z.domain_size + 8) // (t + 1) + 1


@pytest.mark.parametrize("curve_name,circuit", FIXTURES)
@pytest.mark.parametrize("n,t", [(3, 1), (5, 2)])
def test_oracle_shamir_proofs_agree_and_verify(curve_name, circuit, n, t):
    if (n, t) == (5, 2) and circuit == "poseidon" and curve_name == "bls12_381":
        pytest.skip("covered by the other combinations; keeps the CPU suite short")
    curve, z, w, wits, streams = setup(curve_name, circuit, n, t, seed=11)
    proofs, h = orc.prove_shamir(z, n, t, w[:z.n_public + 1], wits, streams, want_h=True)
    for p in proofs[1:]:
        np.testing.assert_array_equal(p, proofs[0])
    assert orc.verify(curve, vk_of(z), w[1:z.n_public + 1], proofs[0])
    # a different sharing / different randomness gives a different but equally valid proof
    curve, z, w, wits2, streams2 = setup(curve_name, circuit, n, t, seed=12)
    other = orc.prove_shamir(z, n, t, w[:z.n_public + 1], wits2, streams2)
    assert not np.array_equal(other[0], proofs[0]) and orc.verify(curve, vk_of(z), w[1:z.n_public + 1], other[0])


@pytest.mark.parametrize("n,t,amount", [(3, 1, None), (3, 1, 100), (5, 2, None)])
def test_oracle_shamir_preprocess(n, t, amount):
    """ShamirProtocol::preprocess (shamir.rs:248): double sharings made up front, in one batch, instead of lazily by 1024"""
    curve, z, w, _, _ = setup("bn254", "poseidon", n, t, seed=13)
    amount = full_amount(z, t) if amount is None else amount
    curve, z, w, wits, streams = setup("bn254", "poseidon", n, t, seed=13, preprocess=amount)
    proofs = orc.prove_shamir(z, n, t, w[:z.n_public + 1], wits, streams, preprocess=amount)
    for p in proofs[1:]:
        np.testing.assert_array_equal(p, proofs[0])
    assert orc.verify(curve, vk_of(z), w[1:z.n_public + 1], proofs[0])
    lazy = orc.prove_shamir(z, n, t, w[:z.n_public + 1], wits, streams)
    assert not np.array_equal(lazy[0], proofs[0])                       # other pairs -> other r, s -> another valid proof


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,circuit,n,t,amount", [("bn254", "poseidon", 3, 1, None), ("bn254", "poseidon", 3, 1, 100), ("bls12_381", "poseidon", 5, 2, None),
                                                           ("bn254", "multiplier2", 9, 4, None), ("bn254", "multiplier2", 9, 4, 3)])
def test_gpu_shamir_preprocess_matches_oracle(curve_name, circuit, n, t, amount):
    """double sharings generated on the GPU (one strided linear combination per share / Vandermonde row; 9 parties = more than 8 terms)
    equal the oracle's buffer_triples(amount); a short amount falls back to the lazy host batches afterwards"""
    ensure_built()
    _, z, _, _, _ = setup(curve_name, circuit, n, t, seed=23)
    amount = full_amount(z, t) if amount is None else amount
    curve, z, w, wits, streams = setup(curve_name, circuit, n, t, seed=23, preprocess=amount)
    want, want_h = orc.prove_shamir(z, n, t, w[:z.n_public + 1], wits, streams, want_h=True, preprocess=amount)
    got, got_h = cg.prove_shamir(curve, fx(curve_name, circuit, "circuit.zkey"), n, t, w[:z.n_public + 1], wits, streams, want_h=True, preprocess=amount)
    np.testing.assert_array_equal(got_h, want_h)
    np.testing.assert_array_equal(got, want)
    assert orc.verify(curve, vk_of(z), w[1:z.n_public + 1], got[0])


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,circuit", FIXTURES)
def test_gpu_shamir_parties_match_oracle(curve_name, circuit):
    ensure_built()
    n, t = 3, 1
    curve, z, w, wits, streams = setup(curve_name, circuit, n, t, seed=21)
    want, want_h = orc.prove_shamir(z, n, t, w[:z.n_public + 1], wits, streams, want_h=True)
    got, got_h = cg.prove_shamir(curve, fx(curve_name, circuit, "circuit.zkey"), n, t, w[:z.n_public + 1], wits, streams, want_h=True)
    np.testing.assert_array_equal(got_h, want_h)
    np.testing.assert_array_equal(got, want)
    assert orc.verify(curve, vk_of(z), w[1:z.n_public + 1], got[0])


@pytest.mark.gpu
def test_gpu_shamir_five_parties_threshold_two():
    ensure_built()
    n, t = 5, 2
    curve, z, w, wits, streams = setup("bn254", "poseidon", n, t, seed=31)
    want = orc.prove_shamir(z, n, t, w[:z.n_public + 1], wits, streams)
    got = cg.prove_shamir(curve, fx("bn254", "poseidon", "circuit.zkey"), n, t, w[:z.n_public + 1], wits, streams)
    np.testing.assert_array_equal(got, want)
    assert orc.verify(curve, vk_of(z), w[1:z.n_public + 1], got[0])
