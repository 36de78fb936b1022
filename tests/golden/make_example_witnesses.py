#!/usr/bin/env python3
"""Fixtures from the reference's EXAMPLE directory and its criterion bench (data only; run in the build container, where /root/reference exists):

    /root/reference/co-circom/co-circom/examples/groth16/test_vectors/{kyc/bn254, kyc/bls12, poseidon, sum_arrays, multiplier2}
    /root/reference/test_vectors/benches/poseidon_hash2/bn254/groth16/poseidon.zkey      (byte-identical to examples/.../poseidon/poseidon.zkey)

into tests/golden/groth16/<curve>/<circuit>/{circuit.zkey, verification_key.json, witness.wtns}, and the PLONK keys of the same example circuits
(co-circom/co-circom/examples/plonk/test_vectors/{kyc/bn254, kyc/bls12, multiplier2, sum_arrays}: zkey + verification key as shipped, the
circuit's witness as above — a circom witness does not depend on the proof system) into tests/golden/plonk/<curve>/<circuit>/.  The zkey and verification key are copied
as shipped.  A witness ships only for kyc/bls12 and poseidon; for the others it is DERIVED here from the shipped .r1cs and the example's own
input.json by propagating the constraints (one unknown wire at a time), and for kyc/bn254 by taking the field-independent small values
(inputs, comparator bits) of the shipped BLS12-381 witness of the same circuit and solving the field-dependent wires (the IsZero inverses)
from the BN254 constraints.  Every witness — shipped or derived — is checked against ALL constraints of its .r1cs before it is written;
tests/test_example_fixtures.py then requires the oracle's proof on it to verify under the SHIPPED verification key.
"""
import json
import os
import shutil
import struct
import sys

REF = "/root/reference"
EX = os.path.join(REF, "co-circom/co-circom/examples/groth16/test_vectors")
EXP = os.path.join(REF, "co-circom/co-circom/examples/plonk/test_vectors")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "groth16")
OUTP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plonk")


def sections(b, magic):
    assert b[:4] == magic
    ns, = struct.unpack_from("<I", b, 8)
    off, secs = 12, {}
    for _ in range(ns):
        t, = struct.unpack_from("<I", b, off); ln, = struct.unpack_from("<Q", b, off + 4)
        secs[t] = (off + 12, ln); off += 12 + ln
    return secs


def read_r1cs(path):
    """iden3 r1cs container: header (section 1) + constraints (section 2: three linear combinations of (wire, coefficient) per row)"""
    b = open(path, "rb").read()
    secs = sections(b, b"r1cs")
    o, _ = secs[1]
    fs, = struct.unpack_from("<I", b, o); prime = int.from_bytes(b[o + 4:o + 4 + fs], "little"); o += 4 + fs
    n_wires, n_pub_out, n_pub_in, n_prv_in = struct.unpack_from("<IIII", b, o); o += 16 + 8
    n_cons, = struct.unpack_from("<I", b, o)
    o, _ = secs[2]
    cons = []
    for _ in range(n_cons):
        row = []
        for _ in range(3):
            n, = struct.unpack_from("<I", b, o); o += 4
            lc = {}
            for _ in range(n):
                w, = struct.unpack_from("<I", b, o); lc[w] = int.from_bytes(b[o + 4:o + 4 + fs], "little"); o += 4 + fs
            row.append(lc)
        cons.append(row)
    return {"prime": prime, "n_wires": n_wires, "n_pub_out": n_pub_out, "n_pub_in": n_pub_in, "n_prv_in": n_prv_in, "cons": cons, "fs": fs}


def read_wtns(path):
    b = open(path, "rb").read()
    secs = sections(b, b"wtns")
    o, _ = secs[1]
    fs, = struct.unpack_from("<I", b, o); prime = int.from_bytes(b[o + 4:o + 4 + fs], "little"); n, = struct.unpack_from("<I", b, o + 4 + fs)
    o, _ = secs[2]
    return prime, [int.from_bytes(b[o + i * fs:o + (i + 1) * fs], "little") for i in range(n)]


def write_wtns(path, prime, values, fs=32):
    """snarkjs wtns container, version 2: section 1 = field size, prime, count; section 2 = the values (witness.rs:40-110 reads exactly this)"""
    s1 = struct.pack("<I", fs) + prime.to_bytes(fs, "little") + struct.pack("<I", len(values))
    s2 = b"".join(v.to_bytes(fs, "little") for v in values)
    with open(path, "wb") as f:
        f.write(b"wtns" + struct.pack("<II", 2, 2) + struct.pack("<IQ", 1, len(s1)) + s1 + struct.pack("<IQ", 2, len(s2)) + s2)


def lc_eval(lc, w, p):
    return sum(c * w[i] for i, c in lc.items()) % p


def satisfied(r, w):
    p = r["prime"]
    return len(w) == r["n_wires"] and w[0] == 1 and all(lc_eval(a, w, p) * lc_eval(b, w, p) % p == lc_eval(c, w, p) for a, b, c in r["cons"])


def solve(r, known):
    """propagate: a constraint with exactly one unknown wire, appearing in exactly one of its three combinations, determines that wire"""
    p = r["prime"]; w = dict(known); w[0] = 1
    progress = True
    while progress and len(w) < r["n_wires"]:
        progress = False
        for a, b, c in r["cons"]:
            unk = {i for lc in (a, b, c) for i in lc if i not in w}
            if len(unk) != 1:
                continue
            x, = unk
            where = [x in a, x in b, x in c]
            if sum(where) != 1:
                continue
            part = lambda lc: sum(co * w[i] for i, co in lc.items() if i != x) % p
            if where[2]:                                                    # A B = C0 + k x
                val = (part(a) * part(b) - part(c)) * pow(c[x], -1, p) % p
            else:
                lin, other = (a, b) if where[0] else (b, a)
                ov = part(other)
                if ov == 0:
                    continue
                val = (part(c) * pow(ov, -1, p) - part(lin)) * pow(lin[x], -1, p) % p
            w[x] = val; progress = True
    if len(w) != r["n_wires"]:
        raise SystemExit(f"could not determine wires {sorted(set(range(r['n_wires'])) - set(w))}")
    return [w[i] for i in range(r["n_wires"])]


def put(curve, circuit, zkey, vk, prime, witness, origin, out=None):
    d = os.path.join(out or OUT, curve, circuit)
    os.makedirs(d, exist_ok=True)
    shutil.copyfile(zkey, os.path.join(d, "circuit.zkey"))
    shutil.copyfile(vk, os.path.join(d, "verification_key.json"))
    write_wtns(os.path.join(d, "witness.wtns"), prime, witness)
    with open(os.path.join(d, "ORIGIN.json"), "w") as f:
        json.dump(origin, f, indent=1); f.write("\n")
    print(f"{'plonk' if out else 'groth16'} {curve}/{circuit}: {len(witness)} wires, witness {origin['witness']}")


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    # circom's wire order: 1, main's outputs, public inputs, private inputs, then the remaining signals
    # -- multiplier2 (main {public [b]}: wires 1, c, b, a) with the example's input.json: a = 3, b = -11
    r = read_r1cs(f"{EX}/multiplier2/multiplier2.r1cs"); p = r["prime"]
    inp = json.load(open(f"{EX}/multiplier2/input.json"))
    w = solve(r, {2: int(inp["b"]) % p, 3: int(inp["a"]) % p})
    assert satisfied(r, w) and w[1] == int(inp["a"]) * int(inp["b"]) % p
    put("bn254", "multiplier2_example", f"{EX}/multiplier2/multiplier2.zkey", f"{EX}/multiplier2/verification_key.json", p, w,
        {"zkey": "examples/groth16/test_vectors/multiplier2/multiplier2.zkey (as shipped)", "witness": "derived from multiplier2.r1cs + input.json (a = 3, b = -11)"})
    assert open(f"{EXP}/multiplier2/multiplier2.r1cs", "rb").read() == open(f"{EX}/multiplier2/multiplier2.r1cs", "rb").read()
    put("bn254", "multiplier2_example", f"{EXP}/multiplier2/multiplier2.zkey", f"{EXP}/multiplier2/verification_key.json", p, w,
        {"zkey": "examples/plonk/test_vectors/multiplier2/multiplier2.zkey (as shipped)", "witness": "derived from multiplier2.r1cs + input.json (a = 3, b = -11)"}, OUTP)
    # -- sum_arrays (main {public [b, c]} = Main(3)): the optimiser removed every (linear) constraint and the unused a[]: wires 1, b[0..2], c[0..2]
    r = read_r1cs(f"{EX}/sum_arrays/sum_arrays.r1cs"); p = r["prime"]
    inp = json.load(open(f"{EX}/sum_arrays/input.json"))
    assert r["n_wires"] == 7 and len(r["cons"]) == 0
    w = [1] + [int(x) for x in inp["b"]] + [int(x) for x in inp["c"]]
    assert satisfied(r, w)
    put("bn254", "sum_arrays", f"{EX}/sum_arrays/sum_arrays.zkey", f"{EX}/sum_arrays/verification_key.json", p, w,
        {"zkey": "examples/groth16/test_vectors/sum_arrays/sum_arrays.zkey (as shipped; 0 constraints, 6 public inputs, no private wire)",
         "witness": "derived from sum_arrays.r1cs + input.json (every wire is a public input)"})
    assert open(f"{EXP}/sum_arrays/sum_arrays.r1cs", "rb").read() == open(f"{EX}/sum_arrays/sum_arrays.r1cs", "rb").read()
    put("bn254", "sum_arrays", f"{EXP}/sum_arrays/sum_arrays.zkey", f"{EXP}/sum_arrays/verification_key.json", p, w,
        {"zkey": "examples/plonk/test_vectors/sum_arrays/sum_arrays.zkey (as shipped)", "witness": "derived from sum_arrays.r1cs + input.json (every wire is a public input)"}, OUTP)
    # -- poseidon (examples) = the criterion bench's key (tests/benches/poseidon_hash2.rs:175-223): zkey + witness as shipped
    r = read_r1cs(f"{EX}/poseidon/poseidon.r1cs")
    p, w = read_wtns(f"{EX}/poseidon/witness.wtns")
    assert p == r["prime"] and satisfied(r, w)
    assert open(f"{EX}/poseidon/poseidon.zkey", "rb").read() == open(f"{REF}/test_vectors/benches/poseidon_hash2/bn254/groth16/poseidon.zkey", "rb").read()
    put("bn254", "poseidon_hash2", f"{EX}/poseidon/poseidon.zkey", f"{EX}/poseidon/verification_key.json", p, w,
        {"zkey": "examples/groth16/test_vectors/poseidon/poseidon.zkey = test_vectors/benches/poseidon_hash2/bn254/groth16/poseidon.zkey (as shipped, byte-identical)",
         "witness": "examples/groth16/test_vectors/poseidon/witness.wtns (as shipped; satisfies poseidon.r1cs)"})
    # -- kyc on BLS12-381: everything as shipped
    r = read_r1cs(f"{EX}/kyc/bls12/kyc.r1cs")
    p381, w381 = read_wtns(f"{EX}/kyc/bls12/witness.wtns")
    assert p381 == r["prime"] and satisfied(r, w381)
    put("bls12_381", "kyc", f"{EX}/kyc/bls12/kyc.zkey", f"{EX}/kyc/bls12/verification_key.json", p381, w381,
        {"zkey": "examples/groth16/test_vectors/kyc/bls12/kyc.zkey (as shipped)", "witness": "examples/groth16/test_vectors/kyc/bls12/witness.wtns (as shipped; satisfies kyc.r1cs)"})
    assert read_wtns(f"{EXP}/kyc/bls12/witness.wtns") == (p381, w381)
    put("bls12_381", "kyc", f"{EXP}/kyc/bls12/kyc.zkey", f"{EXP}/kyc/bls12/verification_key.json", p381, w381,
        {"zkey": "examples/plonk/test_vectors/kyc/bls12/kyc.zkey (as shipped)", "witness": "examples/plonk/test_vectors/kyc/bls12/witness.wtns (as shipped)"}, OUTP)
    # -- kyc on BN254: no witness ships.  Same circuit, same input.json: the small values of the BLS12-381 witness (inputs, comparator bits) are
    # field-independent; the two IsZero inverses that are not 1 are solved from the BN254 constraints
    r = read_r1cs(f"{EX}/kyc/bn254/kyc.r1cs"); p = r["prime"]
    inp = json.load(open(f"{EX}/kyc/input.json"))
    small = {i: v for i, v in enumerate(w381) if v < 1 << 40}
    assert [small[i] for i in range(1, 7)] == [int(x) for x in inp["blacklist"]] + [int(inp["min_age"]), int(inp["country"]), int(inp["age"])]
    w = solve(r, small)
    assert satisfied(r, w) and sorted(set(range(17)) - set(small)) == [7, 8]
    put("bn254", "kyc", f"{EX}/kyc/bn254/kyc.zkey", f"{EX}/kyc/bn254/verification_key.json", p, w,
        {"zkey": "examples/groth16/test_vectors/kyc/bn254/kyc.zkey (as shipped)",
         "witness": "derived: small (field-independent) wires of the shipped BLS12-381 witness of the same circuit and input.json; wires 7, 8 (IsZero inverses) solved from kyc/bn254/kyc.r1cs"})
    assert open(f"{EXP}/kyc/bn254/kyc.r1cs", "rb").read() == open(f"{EX}/kyc/bn254/kyc.r1cs", "rb").read()
    put("bn254", "kyc", f"{EXP}/kyc/bn254/kyc.zkey", f"{EXP}/kyc/bn254/verification_key.json", p, w,
        {"zkey": "examples/plonk/test_vectors/kyc/bn254/kyc.zkey (as shipped)", "witness": "derived as for the Groth16 fixture of the same circuit (tests/golden/groth16/bn254/kyc)"}, OUTP)


if __name__ == "__main__":
    sys.exit(main())
