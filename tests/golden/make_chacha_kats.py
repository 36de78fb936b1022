"""Writes tests/golden/chacha_kats.json: a few F::rand draws over ChaCha12 streams, produced by the ORACLE's restatement (oracle/rngs.hpp) —
regression vectors for the three implementations (oracle, host library, device kernels), NOT reference-produced values: the draw order is
restated from rand_chacha 0.3 / ark-ff 0.4.2 (parity unpinned, see the oracle's header).  Run from the repo root: python tests/golden/make_chacha_kats.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as orc  # noqa: E402

out = {"source": "oracle/rngs.hpp (restated; regression vectors)", "cases": []}
for curve in (orc.BN254, orc.BLS12_381):
    for seed, pos, n in ((bytes(32), 0, 6), (bytes(range(32)), 0, 6), (bytes(range(32)), 13, 5), (bytes(255 - i for i in range(32)), (1 << 36) - 24, 6)):
        vals, after = orc.chacha12_fr_rand(curve, seed, pos, n)
        out["cases"].append({"curve": orc.CURVE_NAMES[curve], "seed": seed.hex(), "word_pos": pos, "n": n, "word_pos_after": after,
                             "draws_montgomery_limbs_le": [["%016x" % int(l) for l in v] for v in vals]})
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "chacha_kats.json"), "w"), indent=1)
print(len(out["cases"]), "cases")
