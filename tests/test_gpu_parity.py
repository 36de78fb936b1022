"""GPU parity tests: every kernel of the hot path, called through the C ABI (libcogroth16_hip.so), against the CPU oracle on
the same seeded inputs.  Bar: bit-exact (integer arithmetic; MSM results compared as affine points, which are unique)."""
import os

import numpy as np
import pytest

import oracle_lib as orc
from oracle_lib import BN254, BLS12_381, FR, FQ, G1, G2
from product import cg, ensure_built

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    ensure_built()
    c = cg.Context(0)
    yield c
    c.close()


def dev(ctx, arr):
    return ctx.to_device(arr)


# ------------------------------------------------------------------------------------------------ pointwise
@pytest.mark.parametrize("curve", [BN254, BLS12_381])
@pytest.mark.parametrize("n", [1, 255, 4099])
def test_vec_ops(ctx, curve, n):
    rng = np.random.default_rng(100 + n)
    a, b, c, d, m = (orc.random_field(curve, FR, n, rng) for _ in range(5))
    # edge values in the first lanes
    if n > 3:
        a[0] = 0; b[1] = 0; a[2] = orc.from_dec(curve, FR, orc.MODULI[(curve, FR)] - 1); b[2] = a[2]
    da, db, dc, dd, dm = (dev(ctx, x) for x in (a, b, c, d, m))
    out = ctx.alloc(n * 32)
    for name, fn in (("add", ctx.vec_add), ("sub", ctx.vec_sub), ("mul", ctx.vec_mul)):
        fn(curve, out, da, db, n)
        np.testing.assert_array_equal(out.download((n, 4)), orc.field_op(curve, FR, name, a, b), err_msg=name)
    # REP3 local product aa*ba + aa*bb + ab*ba (+ mask)   (rep3.rs:656-660)
    mul = lambda x, y: orc.field_op(curve, FR, "mul", x, y)
    add = lambda x, y: orc.field_op(curve, FR, "add", x, y)
    want = add(add(mul(a, c), mul(a, d)), mul(b, c))
    ctx.vec_rep3_mul_local(curve, out, da, db, dc, dd, None, n)
    np.testing.assert_array_equal(out.download((n, 4)), want)
    ctx.vec_rep3_mul_local(curve, out, da, db, dc, dd, dm, n)
    np.testing.assert_array_equal(out.download((n, 4)), add(want, m))
    np.testing.assert_array_equal(ctx.vec_rep3_mul_local_host(curve, a, b, c, d, m), add(want, m))
    np.testing.assert_array_equal(ctx.vec_mul_host(curve, a, b), mul(a, b))
    # in-place sub_assign (rep3.rs:672-679)
    ctx.vec_sub(curve, da, da, db, n)
    np.testing.assert_array_equal(da.download((n, 4)), orc.field_op(curve, FR, "sub", a, b))
    # distribute_powers_and_mul_by_const (rep3.rs:681-688)
    g, cst = orc.random_field(curve, FR, 2, rng)
    ctx.vec_distribute_powers(curve, dc, n, g, cst)
    np.testing.assert_array_equal(dc.download((n, 4)), orc.distribute_powers(curve, c, g, cst))


# ------------------------------------------------------------------------------------------------ SpMV
def rep3_share(curve, vals, rng):
    a = orc.random_field(curve, FR, vals.shape[0], rng); b = orc.random_field(curve, FR, vals.shape[0], rng)
    c = orc.field_op(curve, FR, "sub", orc.field_op(curve, FR, "sub", vals, a), b)
    return [a, b, c], [c, a, b]


def spmv_expected(curve, row_ptr, col, coeff, pub, party, wa, wb):
    """rep3.rs:690-708 / plain.rs:243-258 evaluated with oracle field ops"""
    n_rows = len(row_ptr) - 1
    ninp = pub.shape[0]
    out_a = np.zeros((n_rows, 4), dtype=np.uint64); out_b = np.zeros((n_rows, 4), dtype=np.uint64)
    mul = lambda x, y: orc.field_op(curve, FR, "mul", x, y)
    add = lambda x, y: orc.field_op(curve, FR, "add", x, y)
    for r in range(n_rows):
        for k in range(row_ptr[r], row_ptr[r + 1]):
            idx = int(col[k])
            if idx < ninp:
                t = mul(coeff[k], pub[idx])
                if party <= 0: out_a[r] = add(out_a[r], t)
                elif party == 1: out_b[r] = add(out_b[r], t)
            else:
                out_a[r] = add(out_a[r], mul(coeff[k], wa[idx - ninp]))
                if wb is not None: out_b[r] = add(out_b[r], mul(coeff[k], wb[idx - ninp]))
    return out_a, out_b


@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_spmv_poseidon_matrices(ctx, curve_name):
    curve = {"bn254": BN254, "bls12_381": BLS12_381}[curve_name]
    z = orc.ZKey(curve, os.path.join(GOLDEN, "groth16", curve_name, "poseidon", "circuit.zkey"))
    w = orc.read_wtns(curve, os.path.join(GOLDEN, "groth16", curve_name, "poseidon", "witness.wtns"))
    rng = np.random.default_rng(9)
    pub = w[:z.n_public + 1]
    wa, wb = rep3_share(curve, w[z.n_public + 1:], rng)
    for m in (0, 1):
        rp, col, co = z.matrix(m)
        d_rp, d_col, d_co, d_pub = dev(ctx, rp), dev(ctx, col), dev(ctx, co), dev(ctx, pub)
        oa, ob = ctx.alloc(z.num_constraints * 32), ctx.alloc(z.num_constraints * 32)
        for party in (0, 1, 2):
            ctx.spmv_csr(curve, d_rp, d_col, d_co, z.num_constraints, d_pub, z.n_public + 1, party, dev(ctx, wa[party]), dev(ctx, wb[party]), oa, ob)
            ea, eb = spmv_expected(curve, rp, col, co, pub, party, wa[party], wb[party])
            np.testing.assert_array_equal(oa.download((z.num_constraints, 4)), ea)
            np.testing.assert_array_equal(ob.download((z.num_constraints, 4)), eb)
        # single-component driver (plain / Shamir)
        ctx.spmv_csr(curve, d_rp, d_col, d_co, z.num_constraints, d_pub, z.n_public + 1, -1, dev(ctx, w[z.n_public + 1:]), None, oa, None)
        ea, _ = spmv_expected(curve, rp, col, co, pub, -1, w[z.n_public + 1:], None)
        np.testing.assert_array_equal(oa.download((z.num_constraints, 4)), ea)


# ------------------------------------------------------------------------------------------------ NTT
@pytest.mark.parametrize("curve", [BN254, BLS12_381])
@pytest.mark.parametrize("lg", [0, 1, 2, 5, 9, 10, 11, 12, 13, 16])
def test_ntt_matches_oracle(ctx, curve, lg):
    n = 1 << lg
    rng = np.random.default_rng(200 + lg)
    _, roots, _ = orc.roots_of_unity(curve)
    w = roots[lg]
    x = orc.random_field(curve, FR, n, rng); y = orc.random_field(curve, FR, n, rng)
    fx, fy = ctx.ntt(curve, [x, y], w)
    np.testing.assert_array_equal(fx, orc.ntt(curve, x, w)); np.testing.assert_array_equal(fy, orc.ntt(curve, y, w))
    ix, = ctx.ntt(curve, [x], w, inverse=True)
    np.testing.assert_array_equal(ix, orc.ntt(curve, x, w, inverse=True))
    back, = ctx.ntt(curve, [fx], w, inverse=True)
    np.testing.assert_array_equal(back, x)
    # inverse fused with the coset shift: ifft then distribute_powers(g, 1)   (groth16.rs:175-186)
    g = roots[lg + 1]
    cx, = ctx.ntt(curve, [x], w, inverse=True, coset_gen=g)
    np.testing.assert_array_equal(cx, orc.distribute_powers(curve, orc.ntt(curve, x, w, inverse=True), g, orc.from_dec(curve, FR, 1)))


@pytest.mark.parametrize("curve,lg", [(BN254, 20), (BN254, 24), (BLS12_381, 20), (BLS12_381, 22)])
def test_ntt_large_roundtrip_and_linearity(ctx, curve, lg):
    """size-independent properties at BASELINE-scale lengths (2^20; 2^24 = configs[3]): iNTT(NTT(x)) == x, NTT(x+y) == NTT(x)+NTT(y),
    and 64 outputs against the definition; both scalar fields (the reference's e2e matrix proves on both curves, e2e_tests/mod.rs:20-106)"""
    n = 1 << lg
    rng = np.random.default_rng(7)
    _, roots, _ = orc.roots_of_unity(curve)
    x = orc.random_field(curve, FR, n, rng); y = orc.random_field(curve, FR, n, rng)
    dx, dy, ds = dev(ctx, x), dev(ctx, y), ctx.alloc(n * 32)
    ctx.vec_add(curve, ds, dx, dy, n)
    ctx.ntt_dev(curve, [dx, dy, ds], n, roots[lg])
    t = ctx.alloc(n * 32)
    ctx.vec_add(curve, t, dx, dy, n)
    np.testing.assert_array_equal(t.download((n, 4)), ds.download((n, 4)))
    # spot-check 64 output positions against the definition sum_j x_j w^(jk), evaluated by the oracle on a decimated problem:
    # X[k] for k multiple of n/64 equals the size-64 DFT of the 64-way folded input
    acc = x.reshape(n // 64, 64, 4)
    while acc.shape[0] > 1:
        half = acc.shape[0] // 2
        acc = orc.field_op(curve, FR, "add", acc[:half].reshape(-1, 4), acc[half:].reshape(-1, 4)).reshape(half, 64, 4)
    acc = acc[0]
    want = orc.ntt(curve, acc, roots[6])
    got = dx.download((n, 4))[:: n // 64]
    np.testing.assert_array_equal(got, want)
    ctx.ntt_dev(curve, [dx], n, roots[lg], inverse=True)
    np.testing.assert_array_equal(dx.download((n, 4)), x)


# ------------------------------------------------------------------------------------------------ MSM
def make_points(curve, group, n, rng):
    ks = orc.random_field(curve, FR, n, rng)
    return np.stack([orc.generator_mul(curve, group, k) for k in ks]) if n else np.zeros((0, orc.point_words(curve, group)), dtype=np.uint64)


def msm_check(ctx, curve, group, pts, scalars_list, window=0):
    bases = ctx.register_bases(curve, group, pts)
    ctx.set_msm_window(window)
    got = ctx.msm(bases, scalars_list)
    ctx.set_msm_window(0)
    for j, sc in enumerate(scalars_list):
        want = orc.msm(curve, group, pts, sc, threads=8)
        np.testing.assert_array_equal(cg.point_to_affine(curve, group, got[j]), want, err_msg=f"component {j}")
        np.testing.assert_array_equal(orc.jacobian_to_affine(curve, group, got[j]), want)
    bases.release()


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
@pytest.mark.parametrize("group", [G1, G2])
@pytest.mark.parametrize("n", [1, 2, 33, 700])
def test_msm_random(ctx, curve, group, n):
    rng = np.random.default_rng(300 + n + 10 * group + curve)
    pts = make_points(curve, group, n, rng)
    sa, sb = orc.random_field(curve, FR, n, rng), orc.random_field(curve, FR, n, rng)
    msm_check(ctx, curve, group, pts, [sa, sb])


@pytest.mark.parametrize("window", [2, 5, 9, 13, 16])
def test_msm_window_sizes(ctx, window):
    rng = np.random.default_rng(window)
    pts = make_points(BN254, G1, 300, rng)
    msm_check(ctx, BN254, G1, pts, [orc.random_field(BN254, FR, 300, rng)], window=window)


@pytest.mark.parametrize("group", [G1, G2])
def test_msm_edge_cases(ctx, group):
    """infinity bases, zero / one / p-1 scalars, repeated and opposite points (zkey queries contain all of these)"""
    curve = BN254
    rng = np.random.default_rng(5)
    n = 64
    pts = make_points(curve, group, n, rng)
    sc = orc.random_field(curve, FR, n, rng)
    pts[3] = 0; pts[10] = 0                        # infinity
    pts[5] = pts[4]; sc[5] = sc[4]                 # same point, same scalar -> forces the doubling branch in a bucket
    pts[7] = pts[6]; sc[7] = orc.field_op(curve, FR, "sub", np.zeros(4, dtype=np.uint64), sc[6])   # P*s + P*(-s) = 0
    sc[8] = 0
    sc[9] = orc.from_dec(curve, FR, 1)
    sc[11] = orc.from_dec(curve, FR, orc.MODULI[(curve, FR)] - 1)
    sc[12] = orc.from_dec(curve, FR, 2**253)
    msm_check(ctx, curve, group, pts, [sc])
    # all-zero scalars and all-infinity bases give infinity
    bases = ctx.register_bases(curve, group, pts)
    z = ctx.msm(bases, [np.zeros((n, 4), dtype=np.uint64)])
    assert not cg.point_to_affine(curve, group, z[0]).any()
    bases.release()
    bases = ctx.register_bases(curve, group, np.zeros_like(pts))
    z = ctx.msm(bases, [sc])
    assert not cg.point_to_affine(curve, group, z[0]).any()
    # n == 0
    z = ctx.msm(bases, [np.zeros((0, 4), dtype=np.uint64)], n=0)
    assert not cg.point_to_affine(curve, group, z[0]).any()
    bases.release()


def test_msm_arkworks_struct_layout(ctx):
    """bases handed over as arkworks `Affine{x, y, infinity: bool}` records: 72-byte stride, flag at offset 64"""
    curve, n = BN254, 50
    rng = np.random.default_rng(12)
    pts = make_points(curve, G1, n, rng)
    rec = np.zeros((n, 72), dtype=np.uint8)
    rec[:, :64] = pts.view(np.uint8).reshape(n, 64)
    rec[7, 64] = 1            # flagged infinity although coordinates are non-zero
    pts_ref = pts.copy(); pts_ref[7] = 0
    sc = orc.random_field(curve, FR, n, rng)
    bases = ctx.register_bases(curve, G1, rec, stride=72, infinity_offset=64)
    got = ctx.msm(bases, [sc])
    np.testing.assert_array_equal(cg.point_to_affine(curve, G1, got[0]), orc.msm(curve, G1, pts_ref, sc))
    # sub-slice like &query[1 + pub_len..]  (groth16.rs:221)
    got = ctx.msm(bases, [sc[3:]], offset=3, n=n - 3)
    np.testing.assert_array_equal(cg.point_to_affine(curve, G1, got[0]), orc.msm(curve, G1, pts_ref[3:], sc[3:]))
    bases.release()


@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_msm_on_real_zkey_queries(ctx, curve_name):
    """the five MSMs of create_proof_with_assignment (groth16.rs:248-304) on the poseidon fixture, plain witness as scalars"""
    curve = {"bn254": BN254, "bls12_381": BLS12_381}[curve_name]
    z = orc.ZKey(curve, os.path.join(GOLDEN, "groth16", curve_name, "poseidon", "circuit.zkey"))
    w = orc.read_wtns(curve, os.path.join(GOLDEN, "groth16", curve_name, "poseidon", "witness.wtns"))
    aux = w[z.n_public + 1:]
    h = z.witness_map_plain(w)
    for q, group, sc, off in (("h_query", G1, h, 0), ("l_query", G1, aux, 0), ("a_query", G1, aux, 1 + z.n_public),
                              ("b_g1_query", G1, aux, 1 + z.n_public), ("b_g2_query", G2, aux, 1 + z.n_public)):
        pts = z.points(q)
        bases = ctx.register_bases(curve, group, pts)
        got = ctx.msm(bases, [sc], offset=off, n=sc.shape[0])
        np.testing.assert_array_equal(cg.point_to_affine(curve, group, got[0]), orc.msm(curve, group, pts[off:off + sc.shape[0]], sc), err_msg=q)
        bases.release()


def test_msm_medium_and_linearity(ctx):
    """2^14 points against the oracle, then 2^20 points through size-independent properties:
    MSM(P, a) + MSM(P, b) == MSM(P, a+b) and MSM over a table of repeated points == (sum of scalars) * P"""
    curve = BN254
    rng = np.random.default_rng(77)
    n = 1 << 14
    pts = make_points(curve, G1, n, rng)
    sa, sb = orc.random_field(curve, FR, n, rng), orc.random_field(curve, FR, n, rng)
    msm_check(ctx, curve, G1, pts, [sa, sb])
    big = 1 << 20
    reps = big // n
    pts_big = np.tile(pts, (reps, 1))
    a = orc.random_field(curve, FR, big, rng); b = orc.random_field(curve, FR, big, rng)
    bases = ctx.register_bases(curve, G1, pts_big)
    da, db, ds = dev(ctx, a), dev(ctx, b), ctx.alloc(big * 32)
    ctx.vec_add(curve, ds, da, db, big)
    ra, rb, rs = ctx.msm_dev(bases, [da, db, ds], big)
    lhs = cg.point_to_affine(curve, G1, cg.point_add(curve, G1, ra, rb))
    np.testing.assert_array_equal(lhs, cg.point_to_affine(curve, G1, rs))
    # fold the scalars of identical points on the CPU (oracle field adds) and compare with the 2^14-point oracle MSM
    folded = a.reshape(reps, n, 4)[0].copy()
    for r in range(1, reps):
        folded = orc.field_op(curve, FR, "add", folded, a.reshape(reps, n, 4)[r])
    np.testing.assert_array_equal(cg.point_to_affine(curve, G1, ra), orc.msm(curve, G1, pts, folded, threads=8))
    bases.release()


@pytest.mark.parametrize("curve,group,lg", [(BN254, G1, 24), (BN254, G2, 22), (BLS12_381, G1, 22), (BLS12_381, G2, 20)])
def test_msm_at_maximum_table_size(ctx, curve, group, lg):
    """BASELINE configs[3] table size (2^24 points; G2 at the 2^22 of configs[2]): synthetic table [(1 + i) G], so the exact answer is
    one generator multiplication by sum_i s_i (1 + i), evaluated with oracle field arithmetic.  Precomputed tables (window 20, 13
    windows, 2^24-point index range fully used) and the plain per-window path on a sub-slice; linearity across the two components."""
    import bench_check as bc
    n = 1 << lg
    rng = np.random.default_rng(2400 + lg)
    a = orc.random_field(curve, FR, n, rng); b = orc.random_field(curve, FR, n, rng)
    want = [bc.synth_table_msm(curve, group, s, 1) for s in (a, b)]
    bases = ctx.synth_bases(curve, group, 1, n)
    da, db = dev(ctx, a), dev(ctx, b)
    # per-window bucket sets on the last quarter of the table (offset + length as a caller slices a zkey query)
    off, m = 3 * n // 4, n // 4
    got = ctx.msm_dev(bases, [da.ptr + off * 32, db.ptr + off * 32], m, offset=off)
    part = [bc.synth_table_msm(curve, group, s[off:], 1 + off) for s in (a, b)]
    for j in range(2):
        np.testing.assert_array_equal(cg.point_to_affine(curve, group, got[j]), part[j])
    ctx.precompute_bases(bases, 0)
    got = ctx.msm_dev(bases, [da, db], n)
    for j in range(2):
        np.testing.assert_array_equal(cg.point_to_affine(curve, group, got[j]), want[j])
    ds = ctx.alloc(n * 32); ctx.vec_add(curve, ds, da, db, n)
    rs, = ctx.msm_dev(bases, [ds], n)
    np.testing.assert_array_equal(cg.point_to_affine(curve, group, rs), cg.point_to_affine(curve, group, cg.point_add(curve, group, got[0], got[1])))
    bases.release()


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
@pytest.mark.parametrize("group", [G1, G2])
def test_msm_witness_like_scalars_at_scale(ctx, curve, group):
    """what the plain driver multiplies on a real circuit: half the scalars 0, a third 1, some bytes, the rest full width, 2^20 of
    them — one bucket (digit 1 of the lowest window) holds a third of all entries.  Exact value through the synthetic table
    [(1 + i) G]; classic per-window bucket sets and precomputed window tables (one shared bucket set)."""
    import bench_check as bc
    n = 1 << 20
    rng = np.random.default_rng(77)
    sc = orc.random_field(curve, FR, n, rng)
    sel = rng.random(n)
    small = np.stack([orc.from_dec(curve, FR, v) for v in range(256)])
    sc[sel < 0.5] = 0
    sc[(sel >= 0.5) & (sel < 0.83)] = small[1]
    m = (sel >= 0.83) & (sel < 0.93); sc[m] = small[rng.integers(2, 256, size=int(m.sum()))]
    want = bc.synth_table_msm(curve, group, sc, 1)
    bases = ctx.synth_bases(curve, group, 1, n)
    d = dev(ctx, sc)
    got, = ctx.msm_dev(bases, [d], n)
    np.testing.assert_array_equal(cg.point_to_affine(curve, group, got), want)
    ctx.precompute_bases(bases, 0)
    got, = ctx.msm_dev(bases, [d], n)
    np.testing.assert_array_equal(cg.point_to_affine(curve, group, got), want)
    bases.release()


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
def test_ntt_coset_pair_matches_the_three_reference_steps(ctx, curve):
    """cg_ntt_coset_pair_dev = ifft_in_place; distribute_powers_and_mul_by_const(g, 1); fft_in_place (groth16.rs:175-188) without the
    permutation passes in the middle (inverse passes natural -> bit-reversed, decimation-in-time passes bit-reversed -> natural): exact
    against the oracle for every size 2^0 .. 2^16 (one to three passes per direction, odd and even stage counts), one and two vectors"""
    rng = np.random.default_rng(61)
    _, roots, _ = orc.roots_of_unity(curve)
    one = orc.from_dec(curve, FR, 1)
    for lg in range(0, 17):
        n = 1 << lg
        x, y = orc.random_field(curve, FR, n, rng), orc.random_field(curve, FR, n, rng)
        if n >= 4: x[1] = 0; x[2] = orc.from_dec(curve, FR, orc.MODULI[(curve, FR)] - 1)
        w, g = roots[lg], roots[lg + 1]
        want = [orc.ntt(curve, orc.distribute_powers(curve, orc.ntt(curve, v, w, inverse=True), g, one), w) for v in (x, y)]
        dx, dy = dev(ctx, x), dev(ctx, y)
        ctx.ntt_coset_pair_dev(curve, [dx, dy], n, w, g)
        np.testing.assert_array_equal(dx.download((n, 4)), want[0], err_msg=f"2^{lg}")
        np.testing.assert_array_equal(dy.download((n, 4)), want[1], err_msg=f"2^{lg}")
        dz = dev(ctx, y)
        ctx.ntt_coset_pair_dev(curve, [dz], n, w, g)
        np.testing.assert_array_equal(dz.download((n, 4)), want[1], err_msg=f"2^{lg} single")


@pytest.mark.parametrize("curve,lg", [(BN254, 20), (BN254, 24), (BLS12_381, 20), (BLS12_381, 22)])
def test_ntt_coset_pair_at_scale_equals_the_two_transforms(ctx, curve, lg):
    """2^20 / 2^24 (BASELINE configs[3] size): the pair equals the inverse-with-coset transform followed by the forward transform, which
    the round-trip / linearity / decimated-DFT tests above pin"""
    n = 1 << lg
    rng = np.random.default_rng(lg)
    _, roots, _ = orc.roots_of_unity(curve)
    x = orc.random_field(curve, FR, n, rng)
    a, b = dev(ctx, x), dev(ctx, x)
    ctx.ntt_dev(curve, [a], n, roots[lg], inverse=True, coset_gen=roots[lg + 1]); ctx.ntt_dev(curve, [a], n, roots[lg])
    ctx.ntt_coset_pair_dev(curve, [b], n, roots[lg], roots[lg + 1])
    np.testing.assert_array_equal(b.download((n, 4)), a.download((n, 4)))


def test_msm_async_tickets(ctx):
    curve = BN254
    rng = np.random.default_rng(4)
    n = 500
    p1, p2 = make_points(curve, G1, n, rng), make_points(curve, G2, n, rng)
    s = orc.random_field(curve, FR, n, rng)
    b1, b2 = ctx.register_bases(curve, G1, p1), ctx.register_bases(curve, G2, p2)
    ds = dev(ctx, s)
    # NOTE: tickets share the context's scratch arena in stream order, so begin/begin/end/end is legal
    t1 = ctx.msm_dev_begin(b1, [ds], n)
    t2 = ctx.msm_dev_begin(b2, [ds], n)
    r2 = ctx.msm_end(t2); r1 = ctx.msm_end(t1)
    np.testing.assert_array_equal(cg.point_to_affine(curve, G1, r1[0]), orc.msm(curve, G1, p1, s))
    np.testing.assert_array_equal(cg.point_to_affine(curve, G2, r2[0]), orc.msm(curve, G2, p2, s))
    b1.release(); b2.release()


def test_msm_multi_table_shared_schedule(ctx):
    """cg_msm_dev_begin_multi: l/a/b1 (G1) and b2 (G2) tables times the same two share vectors (groth16.rs:251,267,284,298)"""
    curve, n = BN254, 1500
    rng = np.random.default_rng(41)
    tabs = [(G1, make_points(curve, G1, n + 3, rng)), (G1, make_points(curve, G1, n, rng)), (G2, make_points(curve, G2, n + 1, rng))]
    offs = [3, 0, 1]
    sa, sb = orc.random_field(curve, FR, n, rng), orc.random_field(curve, FR, n, rng)
    bases = [ctx.register_bases(curve, g, p) for g, p in tabs]
    tickets = ctx.msm_dev_begin_multi(bases, [dev(ctx, sa), dev(ctx, sb)], n, offsets=offs)
    for (g, p), o, t in zip(tabs, offs, tickets):
        got = ctx.msm_end(t)
        for j, sc in enumerate((sa, sb)):
            np.testing.assert_array_equal(cg.point_to_affine(curve, g, got[j]), orc.msm(curve, g, p[o:o + n], sc, threads=8))
    for b in bases:
        b.release()


@pytest.mark.parametrize("order,g2_after", [(0, -1), (1, -1), (2, -1), (2, 0), (2, 2), (2, 5)])
@pytest.mark.parametrize("batch", [0, 1, 2, 3])
def test_msm_launch_orders_and_reduction_batches(order, g2_after, batch):
    """the per-context option table (cg_ctx_set_option): every launch order of the (table, component) pairs of a multi-table call —
    caller's order, serpentine, G1 pairs first with the G2 pairs inserted after `g2_after` of them — and every reduction batching
    (each set on its own, per component, per call and field, everything at the end of the call) gives the oracle's sums; with
    precomputed window tables (one shared bucket set) and without; two and three share components (three: the orders fall back to the
    nested loops, the schedule slots are recycled)"""
    curve, n = BN254, 1300
    rng = np.random.default_rng(1000 + 10 * order + batch + g2_after)
    c2 = cg.Context(0)
    try:
        c2.set_option(cg.OPT_MSM_TABLE_ORDER, order); c2.set_option(cg.OPT_MSM_G2_AFTER, g2_after); c2.set_option(cg.OPT_MSM_REDUCE_BATCH, batch)
        assert (c2.get_option(cg.OPT_MSM_TABLE_ORDER), c2.get_option(cg.OPT_MSM_G2_AFTER), c2.get_option(cg.OPT_MSM_REDUCE_BATCH)) == (order, g2_after, batch)
        if batch in (1, 3):                       # two-component calls of this size take the wide path (all sets of a field in one launch) by default: off here, so that the orders above run
            c2.set_option(cg.OPT_MSM_WIDE_SMALL, 0); assert c2.get_option(cg.OPT_MSM_WIDE_SMALL) == 0
        else:
            assert c2.get_option(cg.OPT_MSM_WIDE_SMALL) == 22
        tabs = [(G1, make_points(curve, G1, n, rng)), (G1, make_points(curve, G1, n + 2, rng)), (G2, make_points(curve, G2, n, rng)), (G1, make_points(curve, G1, n, rng))]
        offs = [0, 2, 0, 0]
        sc = [orc.random_field(curve, FR, n, rng) for _ in range(3)]
        bases = [c2.register_bases(curve, g, p) for g, p in tabs]
        for pre in (0, 13):
            if pre:
                for b in bases: c2.precompute_bases(b, pre)
            for k in (2, 3):
                tickets = c2.msm_dev_begin_multi(bases, [c2.to_device(x) for x in sc[:k]], n, offsets=offs)
                for (g, p), o, t in zip(tabs, offs, tickets):
                    got = c2.msm_end(t)
                    for j in range(k):
                        np.testing.assert_array_equal(cg.point_to_affine(curve, g, got[j]), orc.msm(curve, g, p[o:o + n], sc[j], threads=8), err_msg=f"pre {pre} k {k} component {j}")
        for b in bases:
            b.release()
        with pytest.raises(cg.BackendError):
            c2.set_option(cg.OPT_MSM_TABLE_ORDER, 7)
        with pytest.raises(cg.BackendError):
            c2.set_option(99, 1)
    finally:
        c2.close()


@pytest.mark.parametrize("group", [G1, G2])
def test_msm_skewed_scalars(ctx, group):
    """non-uniform digits: thousands of entries in a handful of buckets, so one bucket spans many work chunks
    (plain-driver witnesses look like this: many 0 / 1 / small values)"""
    curve, n = BN254, 6000
    rng = np.random.default_rng(8)
    pts = make_points(curve, group, 64, rng)
    pts = np.tile(pts, (n // 64 + 1, 1))[:n]
    small = np.zeros((n, 4), dtype=np.uint64)
    vals = rng.integers(0, 4, size=n)                    # scalars in {0,1,2,3}
    table = [orc.from_dec(curve, FR, v) for v in range(4)]
    for i in range(n):
        small[i] = table[vals[i]]
    same = np.tile(orc.random_field(curve, FR, 1, rng), (n, 1))   # every scalar identical: every window has ONE busy bucket
    msm_check(ctx, curve, group, pts, [small, same])
    msm_check(ctx, curve, group, pts, [same], window=7)


@pytest.mark.parametrize("curve,c", [(BN254, 8), (BN254, 13), (BN254, 17), (BN254, 18), (BN254, 19), (BN254, 20), (BN254, 21), (BN254, 22), (BLS12_381, 18), (BLS12_381, 20)])
def test_msm_precomputed_windows(ctx, curve, c):
    """cg_bases_precompute: per-window tables 2^(c*j) P_i, one bucket set for all windows; results unchanged.  c <= 17: per-bit sums of the
    shared bucket set (k_msm_bitsum_*); c >= 18: row / column sums + per-bit sums (k_msm_grid_*) over 2^17 .. 2^21 mostly empty buckets"""
    n = 900
    rng = np.random.default_rng(50 + c)
    for group in (G1, G2):
        pts = make_points(curve, group, n + 2, rng)
        pts[5] = 0                                           # an infinity base stays infinity in every window table
        sa, sb = orc.random_field(curve, FR, n, rng), orc.random_field(curve, FR, n, rng)
        sa[0] = 0; sa[1] = orc.from_dec(curve, FR, 1); sa[2] = orc.from_dec(curve, FR, orc.MODULI[(curve, FR)] - 1)
        bases = ctx.register_bases(curve, group, pts)
        ctx.precompute_bases(bases, c)
        got = ctx.msm_dev(bases, [dev(ctx, sa), dev(ctx, sb)], n, offset=2)
        for j, sc in enumerate((sa, sb)):
            np.testing.assert_array_equal(cg.point_to_affine(curve, group, got[j]), orc.msm(curve, group, pts[2:2 + n], sc, threads=8))
        bases.release()


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
@pytest.mark.parametrize("group", [G1, G2])
def test_msm_grid_reduction_dense_buckets(ctx, curve, group):
    """the row / column reduction with every bucket populated: 2^18 points of the synthetic table [(1 + i) G] against per-window tables
    with c = 18 (2^17 buckets, ~30 entries each); exact value = one generator multiplication by sum s_i (1 + i)"""
    import bench_check as bc
    n = 1 << 18
    rng = np.random.default_rng(77)
    bases = ctx.synth_bases(curve, group, 1, n)
    ctx.precompute_bases(bases, 18)
    sa, sb = orc.random_field(curve, FR, n, rng), orc.random_field(curve, FR, n, rng)
    got = ctx.msm_dev(bases, [dev(ctx, sa), dev(ctx, sb)], n)
    for j, sc in enumerate((sa, sb)):
        np.testing.assert_array_equal(cg.point_to_affine(curve, group, got[j]), bc.synth_table_msm(curve, group, sc, 1))
    bases.release()


def test_msm_precomputed_multi_and_skew(ctx):
    curve, n = BN254, 3000
    rng = np.random.default_rng(61)
    t1, t2 = make_points(curve, G1, n, rng), make_points(curve, G2, n + 1, rng)
    b1, b2 = ctx.register_bases(curve, G1, t1), ctx.register_bases(curve, G2, t2)
    ctx.precompute_bases(b1, 16); ctx.precompute_bases(b2, 16)
    same = np.tile(orc.random_field(curve, FR, 1, rng), (n, 1))       # every digit of every scalar identical: maximally skewed buckets
    uni = orc.random_field(curve, FR, n, rng)
    tickets = ctx.msm_dev_begin_multi([b1, b2], [dev(ctx, same), dev(ctx, uni)], n, offsets=[0, 1])
    for (g, p, o), t in zip(((G1, t1, 0), (G2, t2, 1)), tickets):
        got = ctx.msm_end(t)
        for j, sc in enumerate((same, uni)):
            np.testing.assert_array_equal(cg.point_to_affine(curve, g, got[j]), orc.msm(curve, g, p[o:o + n], sc, threads=8))
    # tables with different windows (the automatic choice differs by group and size) or without precomputed tables may share a call:
    # one schedule per window, tickets in the caller's order
    b3 = ctx.register_bases(curve, G1, t1)
    b4 = ctx.register_bases(curve, G2, t2); ctx.precompute_bases(b4, 13)
    tickets = ctx.msm_dev_begin_multi([b3, b1, b4, b2], [dev(ctx, uni)], n, offsets=[0, 0, 1, 1])
    for (g, p, o), t in zip(((G1, t1, 0), (G1, t1, 0), (G2, t2, 1), (G2, t2, 1)), tickets):
        got = ctx.msm_end(t)
        np.testing.assert_array_equal(cg.point_to_affine(curve, g, got[0]), orc.msm(curve, g, p[o:o + n], uni, threads=8))
    for b in (b1, b2, b3, b4):
        b.release()


@pytest.mark.parametrize("cap", [-1, 4, 64, 0])
def test_msm_optimistic_scatter_and_fallback(ctx, cap):
    """one-pass fixed-capacity scatter: cap=4 overflows everywhere and must fall back to the exact schedule; -1 = always exact"""
    curve, n = BN254, 2500
    rng = np.random.default_rng(70)
    p1, p2 = make_points(curve, G1, n, rng), make_points(curve, G2, n, rng)
    b1, b2 = ctx.register_bases(curve, G1, p1), ctx.register_bases(curve, G2, p2)
    uni = orc.random_field(curve, FR, n, rng)
    skew = np.tile(orc.random_field(curve, FR, 1, rng), (n, 1))
    ctx.set_scatter_capacity(cap)
    try:
        tickets = ctx.msm_dev_begin_multi([b1, b2], [dev(ctx, uni), dev(ctx, skew)], n)
        for (g, p), t in zip(((G1, p1), (G2, p2)), tickets):
            got = ctx.msm_end(t)
            for j, sc in enumerate((uni, skew)):
                np.testing.assert_array_equal(cg.point_to_affine(curve, g, got[j]), orc.msm(curve, g, p, sc, threads=8))
        ctx.precompute_bases(b1, 12)
        got = ctx.msm_dev(b1, [dev(ctx, skew), dev(ctx, uni)], n)
        for j, sc in enumerate((skew, uni)):
            np.testing.assert_array_equal(cg.point_to_affine(curve, G1, got[j]), orc.msm(curve, G1, p1, sc, threads=8))
    finally:
        ctx.set_scatter_capacity(-1)
    b1.release(); b2.release()


@pytest.mark.parametrize("curve_name", ["bn254", "bls12_381"])
def test_bases_on_curve_check(ctx, curve_name):
    """device-side point validation (the zkey parser's is_on_curve, circom-types/src/traits.rs:118-123,148-153)"""
    curve = {"bn254": BN254, "bls12_381": BLS12_381}[curve_name]
    z = orc.ZKey(curve, os.path.join(GOLDEN, "groth16", curve_name, "poseidon", "circuit.zkey"))
    for q, group in (("a_query", G1), ("h_query", G1), ("b_g2_query", G2)):
        pts = z.points(q)                                    # real zkey tables (contain infinity records): all valid
        bases = ctx.register_bases(curve, group, pts)
        assert ctx.check_on_curve(bases) == (0, None)
        bases.release()
        bad = pts.copy()
        k = 7 if q != "b_g2_query" else 100
        while not bad[k].any(): k += 1
        bad[k][0] ^= np.uint64(1)                            # flip one bit of an x coordinate
        bases = ctx.register_bases(curve, group, bad)
        nbad, first = ctx.check_on_curve(bases)
        assert (nbad, first) == (1, k)
        assert orc.on_curve(curve, group, pts[k]) and not orc.on_curve(curve, group, bad[k])
        bases.release()


# ---- points on the curve but outside the prime-order subgroup (Python integers; both base fields are = 3 mod 4) --------------------
_Q = {BN254: 21888242871839275222246405745257275088696311157297823662689037894645226208583,
      BLS12_381: 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab}
_R = {BN254: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
      BLS12_381: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001}


def _fp_sqrt(a, p):
    s = pow(a, (p + 1) // 4, p)
    return s if s * s % p == a % p else None


def _fp2_mul(a, b, p): return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def _fp2_sqrt(a, p):
    a0, a1 = a
    if a1 == 0:
        s = _fp_sqrt(a0, p)
        if s is not None: return (s, 0)
        s = _fp_sqrt(-a0 % p, p)
        return (0, s) if s is not None else None
    n = _fp_sqrt((a0 * a0 + a1 * a1) % p, p)
    if n is None: return None
    inv2 = pow(2, -1, p)
    for t in ((a0 + n) * inv2 % p, (a0 - n) * inv2 % p):
        x0 = _fp_sqrt(t, p)
        if x0:
            x = (x0, a1 * pow(2 * x0, -1, p) % p)
            if _fp2_mul(x, x, p) == (a0 % p, a1 % p): return x
    return None


def off_subgroup_point(curve, group, skip=0):
    """an affine point satisfying the curve equation, found by incrementing x (the skip-th one found); lies outside the r-torsion with
    overwhelming probability"""
    p = _Q[curve]
    mont = lambda v: orc.from_dec(curve, FQ, str(v % p))
    if group == G1:
        b = 3 if curve == BN254 else 4
        for x in range(1, 100 + 4 * skip):
            y = _fp_sqrt((x * x * x + b) % p, p)
            if y is not None:
                if skip: skip -= 1; continue
                return np.concatenate([mont(x), mont(y)])
    else:
        if curve == BN254:
            inv = pow(82, -1, p)                                 # 3 / (9 + u) = 3 (9 - u) / 82
            b = (27 * inv % p, -3 * inv % p)
        else:
            b = (4, 4)
        for k in range(1, 100 + 4 * skip):
            x = (k, 1)
            x3 = _fp2_mul(_fp2_mul(x, x, p), x, p)
            y = _fp2_sqrt(((x3[0] + b[0]) % p, (x3[1] + b[1]) % p), p)
            if y is not None:
                if skip: skip -= 1; continue
                return np.concatenate([mont(x[0]), mont(x[1]), mont(y[0]), mont(y[1])])
    raise AssertionError("no point found")


@pytest.mark.parametrize("curve_name,group", [("bn254", G2), ("bls12_381", G1), ("bls12_381", G2), ("bn254", G1)])
@pytest.mark.parametrize("full", [False, True])
def test_bases_subgroup_check(ctx, curve_name, group, full, lib_option):
    """device-side subgroup validation (is_in_correct_subgroup_assuming_on_curve, circom-types/src/traits.rs:121,151): through the
    curve's endomorphisms (csrc/subgroup.hpp, the default) and as [r]P (cg_set_option CG_GOPT_SUBGROUP_FULL)"""
    if full: lib_option(cg.GOPT_SUBGROUP_FULL, 1)
    curve = {"bn254": BN254, "bls12_381": BLS12_381}[curve_name]
    z = orc.ZKey(curve, os.path.join(GOLDEN, "groth16", curve_name, "poseidon", "circuit.zkey"))
    pts = z.points("a_query" if group == G1 else "b_g2_query")
    bases = ctx.register_bases(curve, group, pts)
    assert ctx.check_on_curve(bases) == (0, None) and ctx.check_subgroup(bases) == (0, None)      # real zkey tables pass
    bases.release()
    if curve == BN254 and group == G1:
        return                                                   # cofactor 1: there is no off-subgroup curve point
    bad = off_subgroup_point(curve, group)
    assert orc.on_curve(curve, group, bad)
    r_minus_1 = orc.from_dec(curve, FR, str(_R[curve] - 1))
    rp = orc.point_add(curve, group, orc.points_mul(curve, group, bad[None, :], r_minus_1[None, :])[0], bad)
    assert rp.any()                                              # oracle: [r]P != infinity, i.e. P is outside the subgroup
    k = 57
    tampered = pts.copy(); tampered[k] = bad
    bases = ctx.register_bases(curve, group, tampered)
    assert ctx.check_on_curve(bases) == (0, None)                # still on the curve ...
    assert ctx.check_subgroup(bases) == (1, k)                   # ... but caught by the subgroup pass
    bases.release()
    # a table of nothing but points outside the subgroup, with two valid ones among them: every bad one is counted
    many = np.stack([off_subgroup_point(curve, group, skip=i) for i in range(24)])
    many[5] = pts[3]; many[17] = pts[9]
    bases = ctx.register_bases(curve, group, many)
    assert ctx.check_on_curve(bases) == (0, None) and ctx.check_subgroup(bases) == (22, 0)
    bases.release()


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
@pytest.mark.parametrize("n", [1, 7, 2048, 2049, 5000, 1 << 16, (1 << 18) + 3])
def test_vec_prefix_product_inverse_affine(ctx, curve, n):
    """single-component helpers behind co-plonk's grand product (round2.rs:18-41,146-268) and mul/add_with_public"""
    rng = np.random.default_rng(500 + n)
    x = orc.random_field(curve, FR, n, rng)
    if n > 4:
        x[3] = 0                                                      # a zero in the middle: prefix collapses, inverse keeps 0
    d_x = dev(ctx, x); d_o = ctx.alloc(n * 32)
    # affine: out = c * x + d
    c, d = orc.random_field(curve, FR, 2, rng)
    ctx.vec_affine(curve, d_o, d_x, n, c, d)
    want = orc.field_op(curve, FR, "add", orc.field_op(curve, FR, "mul", x, np.broadcast_to(c, x.shape).copy()), np.broadcast_to(d, x.shape).copy())
    np.testing.assert_array_equal(d_o.download((n, 4)), want)
    # inverse (0 -> 0)
    ctx.vec_inverse(curve, d_o, d_x, n)
    inv = d_o.download((n, 4))
    prod = orc.field_op(curve, FR, "mul", inv, x)
    one = orc.from_dec(curve, FR, "1")
    for i in range(n):
        if x[i].any(): assert np.array_equal(prod[i], one)
        else: assert not inv[i].any()
        if i > 64 and i % 97: continue                                 # spot check beyond the first elements
    if n <= 2049:
        np.testing.assert_array_equal(inv[5 % n], orc.field_inverse(curve, FR, x[5 % n]) if x[5 % n].any() else np.zeros(4, dtype=np.uint64))
    # inclusive prefix product, checked through the recurrence out[i] = out[i-1] * x[i] with oracle multiplications
    ctx.vec_prefix_prod(curve, d_o, d_x, n)
    got = d_o.download((n, 4))
    np.testing.assert_array_equal(got[0], x[0])
    if n > 1:
        np.testing.assert_array_equal(got[1:], orc.field_op(curve, FR, "mul", got[:-1], x[1:]))
    ctx.vec_prefix_sum(curve, d_o, d_x, n)
    got = d_o.download((n, 4))
    np.testing.assert_array_equal(got[0], x[0])
    if n > 1:
        np.testing.assert_array_equal(got[1:], orc.field_op(curve, FR, "add", got[:-1], x[1:]))
    # fill + strided gather
    ctx.vec_fill(curve, d_o, n, c)
    np.testing.assert_array_equal(d_o.download((n, 4)), np.broadcast_to(c, (n, 4)))
    m = n // 4
    if m:
        d_g = ctx.alloc(m * 32)
        ctx.vec_gather_strided(curve, d_g, d_x, m, 1 if n > 4 else 0, 4 if 4 * (m - 1) + 1 < n else 1)
        off, st = (1 if n > 4 else 0), (4 if 4 * (m - 1) + 1 < n else 1)
        np.testing.assert_array_equal(d_g.download((m, 4)), x[off:off + st * m:st][:m])


def test_async_copies_through_pinned_staging(ctx):
    """cg_host_alloc + cg_dev_upload_begin / cg_copy_fence / cg_dev_download_begin / cg_copy_wait: chunks uploaded on the copy stream,
    consumed by a kernel on the context's stream, and streamed back while further work is enqueued"""
    curve = BN254
    rng = np.random.default_rng(99)
    n, ch = 40000, 4096
    a, b = orc.random_field(curve, FR, n, rng), orc.random_field(curve, FR, n, rng)
    pin_a, pin_b, pin_o = ctx.host_alloc((n, 4)), ctx.host_alloc((n, 4)), ctx.host_alloc((n, 4))
    pin_a[:] = a; pin_b[:] = b
    d_a, d_b, d_o = ctx.alloc(n * 32), ctx.alloc(n * 32), ctx.alloc(n * 32)
    tk = None
    for off in range(0, n, ch):
        m = min(ch, n - off)
        ctx.upload_begin(d_a, pin_a[off:off + m], after_stream=False, offset=off * 32)
        tk = ctx.upload_begin(d_b, pin_b[off:off + m], after_stream=False, offset=off * 32)
    ctx.copy_fence(tk)                                            # copies of one direction complete in order
    ctx.vec_mul(curve, d_o, d_a, d_b, n)
    tks = [(off, ctx.download_begin(pin_o[off:off + min(ch, n - off)], d_o, offset=off * 32)) for off in range(0, n, ch)]
    ctx.vec_add(curve, d_a, d_a, d_b, n)                           # enqueued behind the product; the downloads were ordered before it
    for off, t in tks:
        ctx.copy_wait(t)
        np.testing.assert_array_equal(pin_o[off:off + min(ch, n - off)], orc.field_op(curve, FR, "mul", a[off:off + ch], b[off:off + ch]))
    np.testing.assert_array_equal(d_a.download((n, 4)), orc.field_op(curve, FR, "add", a, b))
    # an upload that must wait for enqueued readers of its destination
    pin_a[:] = b
    ctx.vec_add(curve, d_o, d_a, d_b, n)                           # reads d_a (= a + b)
    t = ctx.upload_begin(d_a, pin_a, after_stream=True)
    ctx.copy_fence(t)
    ctx.vec_add(curve, d_b, d_a, d_a, n)                           # sees the uploaded b
    np.testing.assert_array_equal(d_o.download((n, 4)), orc.field_op(curve, FR, "add", orc.field_op(curve, FR, "add", a, b), b))
    np.testing.assert_array_equal(d_b.download((n, 4)), orc.field_op(curve, FR, "add", b, b))
    # downloads ordered behind a MARK of the stream (cg_stream_mark), not behind what is enqueued after it: the product of the mul_vec
    # exchange goes down in chunks while the prover keeps enqueuing transforms
    ctx.vec_mul(curve, d_o, d_a, d_b, n)                           # d_a = b, d_b = 2 b
    mark = ctx.stream_mark()
    for _ in range(20): ctx.vec_add(curve, d_a, d_a, d_b, n)       # later work that must not hold the download back (and does not touch d_o)
    tks = [(off, ctx.download_begin_after(pin_o[off:off + min(ch, n - off)], d_o, mark, offset=off * 32)) for off in range(0, n, ch)]
    want = orc.field_op(curve, FR, "mul", b, orc.field_op(curve, FR, "add", b, b))
    for off, t in tks:
        ctx.copy_wait(t)
        np.testing.assert_array_equal(pin_o[off:off + min(ch, n - off)], want[off:off + ch])
    with pytest.raises(cg.BackendError):
        ctx.download_begin_after(pin_o, d_o, mark + 1000)          # a mark that was never set
    with pytest.raises(cg.BackendError):
        ctx.copy_wait(255)                                         # a ticket that was never issued
    ctx.sync()
    for p in (pin_a, pin_b, pin_o): ctx.host_free(p)


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
def test_vec_lincomb_strided(ctx, curve):
    """strided linear combination (the Shamir share algebra in one launch): forward / negative / interleaving strides, unit coefficients"""
    rng = np.random.default_rng(77)
    n = 5001
    xs = [orc.random_field(curve, FR, 3 * n, rng) for _ in range(3)]
    d_xs = [dev(ctx, x) for x in xs]
    one = orc.from_dec(curve, FR, "1")
    cs = orc.random_field(curve, FR, 3, rng); cs[1] = one
    mulc = lambda c, v: orc.field_op(curve, FR, "mul", v, np.broadcast_to(c, v.shape).copy())
    add = lambda a, b: orc.field_op(curve, FR, "add", a, b)
    # out[2i + 1] = c0 * x0[3i + 2] + 1 * x1[3n - 1 - i] + c2 * x2[i]
    d_o = ctx.alloc(2 * n * 32); ctx.vec_fill(curve, d_o, 2 * n, np.zeros(4, dtype=np.uint64))
    ctx.vec_lincomb(curve, d_o, 1, 2, n, d_xs, [2, 3 * n - 1, 0], [3, -1, 1], cs)
    got = d_o.download((2 * n, 4))
    want = add(add(mulc(cs[0], xs[0][2::3][:n]), xs[1][::-1][:n]), mulc(cs[2], xs[2][:n]))
    np.testing.assert_array_equal(got[1::2], want)
    assert not got[0::2].any()
    # single term, reversed copy with unit coefficient; eight terms
    ctx.vec_lincomb(curve, d_o, 0, 1, n, [d_xs[0]], [n - 1], [-1], one[None])
    np.testing.assert_array_equal(d_o.download((2 * n, 4))[:n], xs[0][:n][::-1])
    c8 = orc.random_field(curve, FR, 8, rng)
    ctx.vec_lincomb(curve, d_o, 0, 1, n, [d_xs[j % 3] for j in range(8)], [j for j in range(8)], [1] * 8, c8)
    want = np.zeros((n, 4), dtype=np.uint64)
    for j in range(8): want = add(want, mulc(c8[j], xs[j % 3][j:j + n]))
    np.testing.assert_array_equal(d_o.download((2 * n, 4))[:n], want)
    with pytest.raises(cg.BackendError):
        ctx.vec_lincomb(curve, d_o, 0, 1, n, [d_xs[0]] * 9, [0] * 9, [1] * 9, orc.random_field(curve, FR, 9, rng))


@pytest.mark.parametrize("group", [G1, G2])
@pytest.mark.parametrize("compact", [True, False])
def test_msm_table_with_many_infinity_points(ctx, group, compact, lib_option):
    """tables that are sparse in points (B queries of real zkeys) are compacted at registration (from 2^14 points on; CG_GOPT_COMPACT_MIN_LOG brings
    this 3000-point table under it) or keep their infinity records: results must not change — whole table, sub-slices, with and without
    precomputed window tables, two tables with the same pattern sharing a call, an all-infinity range"""
    lib_option(cg.GOPT_COMPACT_MIN_LOG, 6 if compact else 20)
    curve = BN254
    rng = np.random.default_rng(808)
    n = 3000
    pts = np.stack([orc.generator_mul(curve, group, s) for s in orc.random_field(curve, FR, 64, rng)])
    table = pts[rng.integers(0, 64, size=n)]
    dead = rng.random(n) < 0.45
    dead[100:400] = True                                            # a long all-infinity stretch
    dead[2100:2102] = True                                          # (the two-offset case below ends on these)
    table[dead] = 0
    sc = [orc.random_field(curve, FR, n, rng), orc.random_field(curve, FR, n, rng)]
    bases = ctx.register_bases(curve, group, table)
    assert ctx.check_on_curve(bases) == (0, None)
    np.testing.assert_array_equal(ctx.bases_download(bases, 0, n), table)      # the original indexing stays visible
    def check(b, lo, cnt):
        got = ctx.msm(b, [s[lo:lo + cnt] for s in sc], offset=lo, n=cnt)
        for j in range(2):
            np.testing.assert_array_equal(cg.point_to_affine(curve, group, got[j]), orc.msm(curve, group, table[lo:lo + cnt], sc[j][lo:lo + cnt]), err_msg=f"range {lo}+{cnt} comp {j}")
    for lo, cnt in ((0, n), (1, n - 1), (57, 1000), (120, 200), (399, 3), (2990, 10)):
        check(bases, lo, cnt)
    ctx.precompute_bases(bases, 12)
    for lo, cnt in ((0, n), (250, 2000), (120, 200)):
        check(bases, lo, cnt)
    # a second table with the SAME infinity pattern and one without infinities in one multi-table call
    other = table.copy(); other[~dead] = pts[rng.integers(0, 64, size=int((~dead).sum()))]
    full = pts[rng.integers(0, 64, size=n)]
    b2 = ctx.register_bases(curve, group, other); b3 = ctx.register_bases(curve, group, full)
    ctx.precompute_bases(b2, 12); ctx.precompute_bases(b3, 12)
    d_sc = [dev(ctx, s) for s in sc]
    tks = ctx.msm_dev_begin_multi([bases, b3, b2], d_sc, n)
    outs = [ctx.msm_end(t) for t in tks]
    for tab, out in zip((table, full, other), outs):
        for j in range(2):
            np.testing.assert_array_equal(cg.point_to_affine(curve, group, out[j]), orc.msm(curve, group, tab, sc[j]))
    # same pattern, DIFFERENT caller offsets that land on the same compacted range (the offsets differ only across infinity entries):
    # each table must still pair point i with scalar i - its own offset
    m = 2000
    assert dead[100] and dead[101] and dead[102 + m - 1] and dead[102 + m - 2]
    tks = ctx.msm_dev_begin_multi([bases, b2], d_sc, m, offsets=[100, 102])
    outs = [ctx.msm_end(t) for t in tks]
    for tab, off, out in zip((table, other), (100, 102), outs):
        for j in range(2):
            np.testing.assert_array_equal(cg.point_to_affine(curve, group, out[j]), orc.msm(curve, group, tab[off:off + m], sc[j][:m]), err_msg=f"offset {off} comp {j}")
    for b in (bases, b2, b3): b.release()


@pytest.mark.gpu
def test_msm_randomised_against_oracle():
    """scripts/fuzz_msm.py for 20 s: random sizes / groups / curves / windows / precomputed tables / scatter capacities / special scalars /
    repeated, opposite and infinity points / sub-slices, every result against the oracle (5159 cases in 150 s when it was written)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_msm.py"), "20", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "agree with the oracle" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("curve_name,group", [("bn254", G2), ("bls12_381", G1), ("bls12_381", G2)])
def test_subgroup_check_fast_and_full_agree_on_mixed_tables(ctx, curve_name, group, lib_option):
    """a table of 384 curve points — multiples of the generator, points outside the subgroup, and sums of both — gets the same verdict
    from the endomorphism-based kernel and from [r]P, and the verdict is the expected one (count and first index)"""
    curve = {"bn254": BN254, "bls12_381": BLS12_381}[curve_name]
    rng = np.random.default_rng(5150)
    gen = cg.point_generator(curve, group)
    offs = [cg.point_from_affine(curve, group, off_subgroup_point(curve, group, skip=i)) for i in range(8)]
    pts, bad = [], []
    for i, k in enumerate(orc.random_field(curve, FR, 384, rng)):
        p = cg.point_scalar_mul(curve, group, gen, k)
        kind = int(rng.integers(0, 4))
        if kind == 0:                                                            # valid + a point outside the subgroup: outside
            p = cg.point_add(curve, group, p, offs[i % 8]); bad.append(i)
        elif kind == 1 and i % 3 == 0:                                          # a small multiple of a point outside the subgroup: outside
            p = cg.point_scalar_mul(curve, group, offs[i % 8], orc.from_dec(curve, FR, str(2 + i % 5))); bad.append(i)
        pts.append(cg.point_to_affine(curve, group, p))
    pts = np.stack(pts)
    want = (len(bad), bad[0])
    for full in (False, True):
        lib_option(cg.GOPT_SUBGROUP_FULL, 1 if full else 0)
        bases = ctx.register_bases(curve, group, pts)
        assert ctx.check_on_curve(bases) == (0, None)
        assert ctx.check_subgroup(bases) == want
        bases.release()
