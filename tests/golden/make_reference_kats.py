#!/usr/bin/env python3
"""Extracts the EXPECTED VALUES (decimal coordinates / field elements) that the reference's own unit tests
hard-code, and stores them as data in tests/golden/reference_kats.json.  Run in the build container only
(/root/reference does not exist on the GPU box); the JSON is committed.

Sources (values only, no code is copied):
  /root/reference/co-circom/circom-types/src/groth16/zkey.rs:336-719   zkey decode KATs (multiplier2, both curves)
  /root/reference/mpc-core/tests/protocols/rep3.rs:242-350             Fr product KAT (rep3_mul_vec_bn)
  /root/reference/co-circom/circom-types/src/witness.rs:101-134        witness KAT
  /root/reference/co-circom/co-plonk/src/round1.rs:346-428             Plonk round-1 commitments [a]_1, [b]_1, [c]_1 (blinding b_i = i)
  /root/reference/co-circom/co-plonk/src/round2.rs:326-355             Plonk round-2 commitment [z]_1
  /root/reference/co-circom/co-plonk/src/round3.rs:553-596             Plonk round-3 commitments [t1]_1, [t2]_1, [t3]_1
  /root/reference/co-circom/co-plonk/src/round4.rs:169-246             Plonk round-4 evaluations
  /root/reference/co-circom/co-plonk/src/round5.rs:391-429             Plonk round-5 opening commitments [Wxi]_1, [Wxiw]_1
  /root/reference/co-circom/co-plonk/src/types.rs:194-227              Keccak256 transcript challenge
"""
import json, re, sys, os

REF = "/root/reference"
out = {}

def let_statements(body):
    """yield (name, text) for each top-level `let name = ...;` in a Rust test body"""
    for m in re.finditer(r"let\s+(?:mut\s+)?(\w+)\s*(?::[^=]+)?=\s*(.*?);\s*\n", body, re.S):
        yield m.group(1), m.group(2)

def tokens(text):
    toks = []
    for m in re.finditer(r'identity\(\)|"(\d+)"', text):
        toks.append("inf" if m.group(0).startswith("identity") else m.group(1))
    return toks

src = open(f"{REF}/co-circom/circom-types/src/groth16/zkey.rs").read()
tests = re.split(r"#\[test\]", src)[1:]
zk = {}
for t in tests:
    name = re.search(r"fn\s+(\w+)", t).group(1)
    path = re.search(r'File::open\("([^"]+)"\)', t)
    entry = {"file": path.group(1).replace("../../test_vectors/", "") if path else None, "values": {}}
    for var, text in let_statements(t):
        tk = tokens(text)
        if tk:
            entry["values"][var] = tk
    if entry["values"]:
        zk[name] = entry
out["zkey"] = zk

src = open(f"{REF}/mpc-core/tests/protocols/rep3.rs").read()
m = re.search(r"fn rep3_mul_vec_bn\(\)(.*?)\n    }\n", src, re.S)
body = m.group(1)
vals = {}
for var, text in let_statements(body):
    tk = re.findall(r'"(\d+)"', text)
    if tk:
        vals[var] = tk
out["rep3_mul_vec_bn"] = vals

src = open(f"{REF}/co-circom/circom-types/src/witness.rs").read()
wt = {}
for t in re.split(r"#\[test\]", src)[1:]:
    name = re.search(r"fn\s+(\w+)", t).group(1)
    path = re.search(r'File::open\("([^"]+)"\)', t)
    nums = re.findall(r'Fr::from\((\d+)\)', t)
    wt[name] = {"file": path.group(1).replace("../../test_vectors/", "") if path else None, "values": nums}
out["witness"] = wt

# snarkjs byte dumps of F.one / G1.one / G2.one in Montgomery LE form (zkey.rs:590-640)
zsrc = open(f"{REF}/co-circom/circom-types/src/groth16/zkey.rs").read()
bufs = {}
for name in ("fq_buf", "g1_buf", "g2_buf"):
    m = re.search(r"fn %s\(\) -> Vec<u8> \{\s*vec!\[(.*?)\]" % name, zsrc, re.S)
    bufs[name] = [int(x) for x in re.findall(r"\d+", m.group(1))]
out["bn254_one_bytes"] = bufs

# Plonk round 1: exact commitments for the deterministic blinding b_i = i (Round1Challenges::deterministic, round1.rs:99-107)
src = open(f"{REF}/co-circom/co-plonk/src/round1.rs").read()
pk = {}
for t in re.split(r"#\[test\]", src)[1:]:
    name = re.search(r"fn\s+(\w+)", t).group(1)
    path = re.search(r'File::open\("([^"]+circuit\.zkey)"\)', t)
    entry = {"file": path.group(1).replace("../../test_vectors/", "") if path else None}
    for which in ("commit_a", "commit_b", "commit_c"):
        m = re.search(r"proof\.%s[^,]*,\s*g1_\w+_from_xy!\(\s*\"(\d+)\",\s*\"(\d+)\"" % which, t, re.S)
        entry[which] = [m.group(1), m.group(2)]
    pk[name] = entry
out["plonk_round1"] = pk

src = open(f"{REF}/co-circom/co-plonk/src/round2.rs").read()
m = re.search(r"commit_z,\s*g1_from_xy!\(\s*\"(\d+)\",\s*\"(\d+)\"", src, re.S)
out["plonk_round2"] = {"test_round2_multiplier2": {"file": "Plonk/bn254/multiplier2/circuit.zkey", "commit_z": [m.group(1), m.group(2)]}}

src = open(f"{REF}/co-circom/co-plonk/src/round3.rs").read()
r3 = {}
for which in ("commit_t1", "commit_t2", "commit_t3"):
    m = re.search(r"proof\.%s,\s*g1_from_xy!\(\s*\"(\d+)\",\s*\"(\d+)\"" % which, src, re.S)
    r3[which] = [m.group(1), m.group(2)]
out["plonk_round3"] = {"test_round3_multiplier2": dict(file="Plonk/bn254/multiplier2/circuit.zkey", **r3)}

src = open(f"{REF}/co-circom/co-plonk/src/round4.rs").read()
r4 = {}
for which in ("eval_a", "eval_b", "eval_c", "eval_zw", "eval_s1", "eval_s2"):
    m = re.search(r"proof\.%s,\s*ark_bn254::Fr::from_str\(\s*\"(\d+)\"" % which, src, re.S)
    r4[which] = m.group(1)
out["plonk_round4"] = {"test_round4_multiplier2": dict(file="Plonk/bn254/multiplier2/circuit.zkey", **r4)}
src = open(f"{REF}/co-circom/co-plonk/src/round5.rs").read()
r5 = {}
for which in ("wxi", "wxiw"):
    m = re.search(r"proof\.%s,\s*g1_from_xy!\(\s*\"(\d+)\",\s*\"(\d+)\"" % which, src, re.S)
    r5[which] = [m.group(1), m.group(2)]
out["plonk_round5"] = {"test_round5_multiplier2": dict(file="Plonk/bn254/multiplier2/circuit.zkey", **r5)}

# transcript KAT: the sequence of add_point / add_scalar calls and the expected challenge (types.rs:194-227)
src = open(f"{REF}/co-circom/co-plonk/src/types.rs").read()
body = src[src.index("fn test_keccak_transcript"):]
items = []
for mm in re.finditer(r'transcript\.add_point\(to_g1_bn254!\(\s*"(\d+)",\s*"(\d+)"\s*\)\)|transcript\.add_point\(ark_bn254::G1Affine::identity\(\)\)|transcript\.add_scalar\(\s*ark_bn254::Fr::from_str\(\s*"(\d+)"', body, re.S):
    if mm.group(1): items.append(["point", mm.group(1), mm.group(2)])
    elif mm.group(3): items.append(["scalar", mm.group(3)])
    else: items.append(["infinity"])
exp = re.search(r'assert_eq!\(\s*ark_bn254::Fr::from_str\(\s*"(\d+)"', body, re.S).group(1)
out["plonk_transcript"] = {"items": items, "challenge": exp}

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
json.dump(out, open(dst, "w"), indent=1)
print({k: (list(v.keys()) if isinstance(v, dict) else len(v)) for k, v in out.items()})
for k, v in zk.items():
    print(k, v["file"], {a: len(b) for a, b in v["values"].items()})
print(out["rep3_mul_vec_bn"].keys(), wt)
