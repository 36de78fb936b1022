#!/usr/bin/env python3
"""Second, independent restatement of the `.shared` witness container (plain Python integers and struct.pack; nothing of the product or
of the oracle is imported).  PARITY STAYS UNPINNED: the reference snapshot holds no `.shared` file and cannot be run here, so this is a
restatement too — but it was written from the reference's types on its own, and the product's C++ reader must parse what it writes.

Layout, from the reference's types:
  co-circom/co-circom/src/bin/co-circom.rs:330,400,449   bincode::serialize_into(file, &SharedWitness)      (bincode 1.x defaults:
                                                          little-endian, fixed-width integers, u64 lengths)
  co-circom/co-circom-snarks/src/lib.rs:24-41             struct SharedWitness { public_inputs, witness }, both fields through
  co-circom/co-circom-snarks/src/serde_compat.rs:5-13     ark_se = serializer.serialize_bytes(ark compressed bytes) = u64 length || bytes
  ark-serialize 0.4, Vec<T>                                u64 element count || elements; Fp = 32 bytes little-endian CANONICAL value
  mpc-core/src/protocols/rep3/fieldshare.rs:232-236        Rep3PrimeFieldShareVec { a: Vec<F>, b: Vec<F> }  (derive: fields in order)
  mpc-core/src/protocols/shamir/fieldshare.rs:152-155      ShamirPrimeFieldShareVec { a: Vec<F> }
The values shared are the reference's own witness KAT [1, 33, 3, 11] (circom-types/src/witness.rs:101-134; multiplier2) and a longer
vector; shares come from Python's Mersenne twister with a fixed seed.  Output: tests/golden/shared/*.shared + expected.json (decimal)."""
import json
import os
import random
import struct

R = {"bn254": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
     "bls12_381": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001}
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shared")


def ark_vec(vals):
    return struct.pack("<Q", len(vals)) + b"".join(int(v).to_bytes(32, "little") for v in vals)


def serde_bytes(b):
    return struct.pack("<Q", len(b)) + b


def shared_witness(public_inputs, share_vectors):
    """share_vectors: [a, b] for REP3, [a] for Shamir"""
    return serde_bytes(ark_vec(public_inputs)) + serde_bytes(b"".join(ark_vec(v) for v in share_vectors))


def main():
    os.makedirs(HERE, exist_ok=True)
    rnd = random.Random(0x5A4ED)
    expected = {}
    for curve, r in R.items():
        for name, full in (("multiplier2", [1, 33, 3, 11]), ("vec37", [1, 7] + [rnd.randrange(r) for _ in range(35)])):
            n_pub = 2                                   # the constant one and one public signal, as in the multiplier2 fixture (n_public = 1)
            pub, priv = full[:n_pub], full[n_pub:]
            # additive REP3 shares x = x0 + x1 + x2; party i holds (a, b) = (x_i, x_(i-1))   (rep3.rs:57-68,124-150)
            x = [[rnd.randrange(r) for _ in priv] for _ in range(2)]
            x.append([(v - s0 - s1) % r for v, s0, s1 in zip(priv, x[0], x[1])])
            for i in range(3):
                a, b = x[i], x[(i + 2) % 3]
                fn = f"{curve}.{name}.rep3.party{i}.shared"
                open(os.path.join(HERE, fn), "wb").write(shared_witness(pub, [a, b]))
                expected[fn] = {"curve": curve, "protocol": "rep3", "public_inputs": [str(v) for v in pub], "a": [str(v) for v in a], "b": [str(v) for v in b],
                                "opens_to": [str(v) for v in priv]}
            # Shamir, 3 parties, threshold 1: share of party p = f(p + 1), f(X) = secret + c1 X   (shamir/shamir_core.rs:8-31)
            c1 = [rnd.randrange(r) for _ in priv]
            for p in range(3):
                a = [(v + c * (p + 1)) % r for v, c in zip(priv, c1)]
                fn = f"{curve}.{name}.shamir.party{p}.shared"
                open(os.path.join(HERE, fn), "wb").write(shared_witness(pub, [a]))
                expected[fn] = {"curve": curve, "protocol": "shamir", "public_inputs": [str(v) for v in pub], "a": [str(v) for v in a]}
    json.dump(expected, open(os.path.join(HERE, "expected.json"), "w"), indent=0)
    print(f"wrote {len(expected)} files to {HERE}")


if __name__ == "__main__":
    main()
