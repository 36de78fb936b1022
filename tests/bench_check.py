"""Oracle evaluation of one bench.py step (test infrastructure): the same party-0 REP3 prove compute, from the inputs bench.py
dumps with --dump-inputs, using only oracle field/NTT/curve primitives.  The synthetic tables are [(first + i) * G], so each MSM
collapses to one generator multiplication by sum_i s_i * (first + i) — cheap at any size."""
import numpy as np

import oracle_lib as orc
from oracle_lib import BN254, FR, G1, G2

TABLE_GROUP = {"h": G1, "l": G1, "a": G1, "b1": G1, "b2": G2}
TABLE_FIRST = {"h": 1, "l": 3, "a": 5, "b1": 7, "b2": 1}     # mirrors bench.TABLE_FIRST (checked by the test)
R_BN254 = 21888242871839275222246405745257275088548364400416034343698204186575808495617

mul = lambda x, y, curve=BN254: orc.field_op(curve, FR, "mul", x, y)
add = lambda x, y, curve=BN254: orc.field_op(curve, FR, "add", x, y)
sub = lambda x, y, curve=BN254: orc.field_op(curve, FR, "sub", x, y)


def to_mont(raw, curve=BN254):
    """(n, 4) canonical limbs -> Montgomery form: mont_mul(x, R^2) = x R   (R = 2^256 for both scalar fields)"""
    r2 = orc.from_dec(curve, FR, str(pow(2, 256, orc.MODULI[(curve, FR)])))   # Montgomery representative of R = raw limbs of R^2 mod r
    return mul(raw, np.broadcast_to(r2, raw.shape).copy(), curve)


def field_sum(x, curve=BN254):
    x = x.copy()
    while x.shape[0] > 1:
        if x.shape[0] & 1:
            x = np.concatenate([x, np.zeros((1, 4), dtype=np.uint64)])
        h = x.shape[0] // 2
        x = add(x[:h], x[h:], curve)
    return x[0]


def synth_table_msm(curve, group, scalars, first):
    """exact MSM value over the synthetic table [(first + i) G], i = 0 .. n-1: one generator multiplication by sum_i s_i (first + i)"""
    n = scalars.shape[0]
    idx = np.zeros((n, 4), dtype=np.uint64); idx[:, 0] = np.arange(n, dtype=np.uint64) + np.uint64(first)
    return orc.generator_mul(curve, group, field_sum(mul(scalars, to_mont(idx, curve), curve), curve))


def spmv_party0(rp, col, co, pub, wa, wb, n_inputs):
    """rep3.rs:690-708 for party 0: public signals go to component a"""
    rp, col = rp.astype(np.int64), col.astype(np.int64)
    n_rows = rp.shape[0] - 1
    xa = np.concatenate([pub, wa]); xb = np.concatenate([np.zeros_like(pub), wb])
    ta, tb = mul(co, xa[col]), mul(co, xb[col])
    out_a = np.zeros((n_rows, 4), dtype=np.uint64); out_b = np.zeros((n_rows, 4), dtype=np.uint64)
    row_of = np.repeat(np.arange(n_rows), np.diff(rp))
    k_in_row = np.arange(col.shape[0]) - rp[row_of]
    for k in range(int(np.diff(rp).max())):                       # rows are short: one vectorised pass per position
        sel = k_in_row == k
        rows = row_of[sel]
        out_a[rows] = add(out_a[rows], ta[sel]); out_b[rows] = add(out_b[rows], tb[sel])
    return out_a, out_b


def coset_eval(v, omega, g):
    """ifft, distribute_powers(g), fft  (groth16.rs:186-215)"""
    one = orc.from_dec(BN254, FR, "1")
    return orc.ntt(BN254, orc.distribute_powers(BN254, orc.ntt(BN254, v, omega, inverse=True), g, one), omega)


def expected_results(d):
    m, nc, n_inputs, n_aux = (int(x) for x in d["in_shape"])
    pad = lambda v: np.concatenate([v, np.zeros((m - v.shape[0], 4), dtype=np.uint64)])
    aa, ab = spmv_party0(d["in_rpA"], d["in_colA"], d["in_coA"], d["in_pub"], d["in_wa"], d["in_wb"], n_inputs)
    ba, bb = spmv_party0(d["in_rpB"], d["in_colB"], d["in_coB"], d["in_pub"], d["in_wa"], d["in_wb"], n_inputs)
    aa, ab, ba, bb = pad(aa), pad(ab), pad(ba), pad(bb)
    aa[nc:nc + n_inputs] = d["in_pub"]
    rep3_local = lambda xa, xb, ya, yb, mask: add(add(mul(xa, add(ya, yb)), mul(xb, ya)), mask)      # rep3.rs:656-660
    ca, cb = rep3_local(aa, ab, ba, bb, d["in_mask1"]), d["in_recv1"]
    om, g = d["in_omega"], d["in_coset_g"]
    aa, ab, ba, bb, ca, cb = (coset_eval(v, om, g) for v in (aa, ab, ba, bb, ca, cb))
    ha = sub(rep3_local(aa, ab, ba, bb, d["in_mask2"]), ca)
    hb = sub(d["in_recv2"], cb)
    out = {}
    for t in ("h", "l", "a", "b1", "b2"):
        sc = (ha, hb) if t == "h" else (d["in_wa"], d["in_wb"])
        n = sc[0].shape[0]
        idx = np.zeros((n, 4), dtype=np.uint64); idx[:, 0] = np.arange(n, dtype=np.uint64) + np.uint64(TABLE_FIRST[t])
        wts = to_mont(idx)
        out[t] = np.stack([orc.generator_mul(BN254, TABLE_GROUP[t], field_sum(mul(s, wts))) for s in sc])
    return out
