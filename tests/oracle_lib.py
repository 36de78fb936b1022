"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE: imported only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  All arrays are numpy uint64 in the product ABI's convention
(little-endian Montgomery limbs; packed affine points, (0,0) = infinity)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

BN254, BLS12_381 = 0, 1
FR, FQ = 0, 1
G1, G2 = 0, 1
CURVE_NAMES = {BN254: "bn254", BLS12_381: "bls12_381"}


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    subprocess.check_call(["g++", "-O3", "-mbmi2", "-madx", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", LIB_PATH,
                           os.path.join(ORACLE_DIR, "oracle_capi.cpp")])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_last_error.restype = C.c_char_p
        _lib.orc_zkey_open.restype = C.c_void_p
        _lib.orc_zkey_open.argtypes = [C.c_int, C.c_char_p]
        _lib.orc_zkey_close.argtypes = [C.c_void_p]
    return _lib


def _chk(rc):
    if rc < 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def nlimbs(curve, which):
    return 6 if (curve == BLS12_381 and which == FQ) else 4


def from_dec(curve, which, s):
    out = np.zeros(nlimbs(curve, which), dtype=np.uint64)
    _chk(lib().orc_from_dec(curve, which, str(s).encode(), _p(out)))
    return out


def to_dec(curve, which, limbs):
    limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
    buf = C.create_string_buffer(256)
    _chk(lib().orc_to_dec(curve, which, _p(limbs), buf, C.c_size_t(256)))
    return buf.value.decode()


def field_op(curve, which, op, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    n = a.size // nlimbs(curve, which)
    _chk(lib().orc_field_op(curve, which, {"add": 0, "sub": 1, "mul": 2}[op], _p(a), _p(b), _p(out), C.c_size_t(n)))
    return out


def field_inverse(curve, which, a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    _chk(lib().orc_field_inverse(curve, which, _p(a), _p(out)))
    return out


def roots_of_unity(curve):
    q = np.zeros(4, dtype=np.uint64)
    roots = np.zeros((40, 4), dtype=np.uint64)
    ta = C.c_int(0)
    _chk(lib().orc_roots_of_unity(curve, _p(q), _p(roots), C.byref(ta)))
    return q, roots[: ta.value + 1].copy(), ta.value


def groth16_domain(curve, pow_, num_constraints, num_inputs):
    omega = np.zeros(4, dtype=np.uint64); g = np.zeros(4, dtype=np.uint64); m = C.c_size_t(0)
    _chk(lib().orc_groth16_domain(curve, C.c_size_t(pow_), C.c_size_t(num_constraints), C.c_size_t(num_inputs), _p(omega), _p(g), C.byref(m)))
    return omega, g, m.value


def ntt(curve, data, omega, inverse=False):
    out = np.ascontiguousarray(data, dtype=np.uint64).copy()
    omega = np.ascontiguousarray(omega, dtype=np.uint64)
    _chk(lib().orc_ntt(curve, _p(out), C.c_size_t(out.size // 4), _p(omega), int(inverse)))
    return out


def dft_naive(curve, data, omega):
    data = np.ascontiguousarray(data, dtype=np.uint64)
    out = np.empty_like(data)
    _chk(lib().orc_dft_naive(curve, _p(data), _p(out), C.c_size_t(data.size // 4), _p(np.ascontiguousarray(omega, dtype=np.uint64))))
    return out


def distribute_powers(curve, data, g, c):
    out = np.ascontiguousarray(data, dtype=np.uint64).copy()
    _chk(lib().orc_distribute_powers(curve, _p(out), C.c_size_t(out.size // 4), _p(np.ascontiguousarray(g)), _p(np.ascontiguousarray(c))))
    return out


def point_words(curve, group):
    return nlimbs(curve, FQ) * (2 if group == G1 else 4)


def msm(curve, group, points, scalars, algo="pippenger", threads=1):
    points = np.ascontiguousarray(points, dtype=np.uint64); scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = scalars.size // 4
    assert points.size == n * point_words(curve, group)
    out = np.zeros(point_words(curve, group), dtype=np.uint64)
    _chk(lib().orc_msm(curve, group, 0 if algo == "pippenger" else 1, _p(points), _p(scalars), C.c_size_t(n), threads, _p(out)))
    return out


def jacobian_to_affine(curve, group, jac):
    jac = np.ascontiguousarray(jac, dtype=np.uint64)
    out = np.zeros(point_words(curve, group), dtype=np.uint64)
    _chk(lib().orc_jacobian_to_affine(curve, group, _p(jac), _p(out)))
    return out


def on_curve(curve, group, pt):
    return bool(_chk(lib().orc_on_curve(curve, group, _p(np.ascontiguousarray(pt, dtype=np.uint64)))))


def generator_mul(curve, group, scalar):
    out = np.zeros(point_words(curve, group), dtype=np.uint64)
    _chk(lib().orc_generator_mul(curve, group, _p(np.ascontiguousarray(scalar, dtype=np.uint64)), _p(out)))
    return out


def points_mul(curve, group, pts, scalars):
    pts = np.ascontiguousarray(pts, dtype=np.uint64); scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.empty_like(pts)
    _chk(lib().orc_points_mul(curve, group, _p(pts), _p(scalars), C.c_size_t(scalars.size // 4), _p(out)))
    return out


def point_add(curve, group, a, b):
    out = np.zeros(point_words(curve, group), dtype=np.uint64)
    _chk(lib().orc_point_add(curve, group, _p(np.ascontiguousarray(a, dtype=np.uint64)), _p(np.ascontiguousarray(b, dtype=np.uint64)), _p(out)))
    return out


class ZKey:
    SEL = {"ic": 0, "a_query": 1, "b_g1_query": 2, "b_g2_query": 3, "l_query": 4, "h_query": 5, "vk_g1": 6, "vk_g2": 7}

    def __init__(self, curve, path):
        self.curve = curve
        self.h = lib().orc_zkey_open(curve, path.encode())
        if not self.h:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())
        info = (C.c_size_t * 7)()
        _chk(lib().orc_zkey_info(C.c_void_p(self.h), info))
        (self.n_vars, self.n_public, self.domain_size, self.pow, self.num_constraints, self.nnz_a, self.nnz_b) = [int(x) for x in info]
        self.n_aux = self.n_vars - self.n_public - 1

    def close(self):
        if self.h:
            lib().orc_zkey_close(C.c_void_p(self.h)); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def points(self, which):
        nq = nlimbs(self.curve, FQ)
        counts = {"ic": (self.n_public + 1, 2), "a_query": (self.n_vars, 2), "b_g1_query": (self.n_vars, 2), "b_g2_query": (self.n_vars, 4),
                  "l_query": (self.n_aux, 2), "h_query": (self.domain_size, 2), "vk_g1": (3, 2), "vk_g2": (3, 4)}
        n, w = counts[which]
        out = np.zeros((n, w * nq), dtype=np.uint64)
        _chk(lib().orc_zkey_points(C.c_void_p(self.h), self.SEL[which], _p(out)))
        return out

    def matrix(self, m):
        nnz = self.nnz_a if m == 0 else self.nnz_b
        row_ptr = np.zeros(self.num_constraints + 1, dtype=np.uint32)
        col = np.zeros(nnz, dtype=np.uint32)
        coeff = np.zeros((nnz, 4), dtype=np.uint64)
        _chk(lib().orc_zkey_matrix(C.c_void_p(self.h), m, _p(row_ptr), _p(col), _p(coeff)))
        return row_ptr, col, coeff

    def witness_map_plain(self, full_witness):
        w = np.ascontiguousarray(full_witness, dtype=np.uint64)
        out = np.zeros((self.domain_size, 4), dtype=np.uint64)
        _chk(lib().orc_witness_map_plain(C.c_void_p(self.h), _p(w), _p(out)))
        return out

    def prove_plain(self, full_witness, r, s, threads=1, timing=False):
        w = np.ascontiguousarray(full_witness, dtype=np.uint64)
        out = np.zeros(8 * nlimbs(self.curve, FQ), dtype=np.uint64)
        secs = C.c_double(0)
        _chk(lib().orc_prove_plain(C.c_void_p(self.h), _p(w), _p(np.ascontiguousarray(r)), _p(np.ascontiguousarray(s)), threads, _p(out), C.byref(secs)))
        return (out, secs.value) if timing else out

    def prove_rep3(self, pub, wit_a, wit_b, streams, threads=1, want_h=False):
        """pub: (n_public+1,4); wit_a/wit_b: lists of 3 arrays (n_aux,4); streams: list of 3 arrays (L,4)."""
        pub = np.ascontiguousarray(pub, dtype=np.uint64)
        wa = [np.ascontiguousarray(x, dtype=np.uint64) for x in wit_a]
        wb = [np.ascontiguousarray(x, dtype=np.uint64) for x in wit_b]
        st = [np.ascontiguousarray(x, dtype=np.uint64) for x in streams]
        arr = lambda xs: (C.c_void_p * 3)(*[x.ctypes.data for x in xs])
        psz = 8 * nlimbs(self.curve, FQ)
        out = np.zeros((3, psz), dtype=np.uint64)
        h = np.zeros((2, self.domain_size, 4), dtype=np.uint64) if want_h else None
        _chk(lib().orc_prove_rep3(C.c_void_p(self.h), _p(pub), arr(wa), arr(wb), arr(st), C.c_size_t(st[0].shape[0]), threads, _p(out), _p(h)))
        return (out, h) if want_h else out


def prove_shamir(self_zkey, n, t, pub, wits, streams, threads=1, want_h=False, preprocess=0):
    """oracle/shamir.hpp: n parties in lock-step; returns (n proofs[, party 0's h shares])"""
    z = self_zkey
    nq = nlimbs(z.curve, FQ)
    keep = [[np.ascontiguousarray(x, dtype=np.uint64) for x in lst] for lst in (wits, streams)]
    arr = lambda lst: (C.c_void_p * n)(*[x.ctypes.data for x in lst])
    out = np.zeros((n, 8 * nq), dtype=np.uint64)
    h = np.zeros((z.domain_size, 4), dtype=np.uint64) if want_h else None
    _chk(lib().orc_prove_shamir(C.c_void_p(z.h), n, t, _p(np.ascontiguousarray(pub, dtype=np.uint64)), arr(keep[0]), arr(keep[1]),
                                C.c_size_t(keep[1][0].shape[0]), C.c_size_t(int(preprocess)), threads, _p(out), _p(h) if want_h else None))
    return (out, h) if want_h else out


def shamir_share(curve, vals, n, t, rng):
    """Shamir shares of `vals` for parties 1..n with random degree-t polynomials (shamir_core.rs:8-31)"""
    coeffs = [random_field(curve, FR, vals.shape[0], rng) for _ in range(t)]
    shares = []
    for p in range(1, n + 1):
        acc = vals.copy(); xp = p
        for c in coeffs:
            x = np.broadcast_to(from_dec(curve, FR, str(xp)), c.shape).copy()
            acc = field_op(curve, FR, "add", acc, field_op(curve, FR, "mul", c, x)); xp *= p
        shares.append(acc)
    return shares


def read_wtns(curve, path):
    n = C.c_size_t(0)
    _chk(lib().orc_wtns_read(curve, path.encode(), None, C.c_size_t(0), C.byref(n)))
    out = np.zeros((n.value, 4), dtype=np.uint64)
    _chk(lib().orc_wtns_read(curve, path.encode(), _p(out), C.c_size_t(n.value), C.byref(n)))
    return out


def verify(curve, vk, pub, proof):
    """vk: dict with alpha1, beta2, gamma2, delta2, ic (packed arrays); pub: (n_pub,4); proof: packed A||B||C."""
    pub = np.ascontiguousarray(pub, dtype=np.uint64).reshape(-1, 4)
    ic = np.ascontiguousarray(vk["ic"], dtype=np.uint64)
    return bool(_chk(lib().orc_verify(curve, _p(np.ascontiguousarray(vk["alpha1"])), _p(np.ascontiguousarray(vk["beta2"])),
                                      _p(np.ascontiguousarray(vk["gamma2"])), _p(np.ascontiguousarray(vk["delta2"])),
                                      _p(ic), C.c_size_t(pub.shape[0]), _p(pub), _p(np.ascontiguousarray(proof, dtype=np.uint64)))))


def pairing(curve, g1_affine, g2_affine):
    """e(P, Q) with snarkjs' / arkworks' value convention (oracle/pairing.hpp::optimal_ate_pairing) as (2, 3, 2, limbs): the JSON layout
    of `vk_alphabeta_12`, Montgomery form"""
    nq = 6 if curve == BLS12_381 else 4
    out = np.zeros((2, 3, 2, nq), dtype=np.uint64)
    _chk(lib().orc_pairing(curve, _p(np.ascontiguousarray(g1_affine, dtype=np.uint64)), _p(np.ascontiguousarray(g2_affine, dtype=np.uint64)), _p(out)))
    return out


def pairing_selfcheck(curve, scalar):
    return bool(_chk(lib().orc_pairing_selfcheck(curve, _p(np.ascontiguousarray(scalar, dtype=np.uint64)))))


def bench_rep3_party(curve, log_m, threads, seed=1):
    """cpu_baseline workload: seconds for one REP3 party's prove compute at m = 2^log_m (+ per-stage seconds)"""
    lib().orc_bench_rep3_party.restype = C.c_double
    stage = (C.c_double * 4)()
    t = lib().orc_bench_rep3_party(curve, int(log_m), int(threads), C.c_uint64(seed), stage)
    if t < 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())
    return t, {"spmv_pointwise_s": stage[0], "ntt_s": stage[1], "msm_g1_s": stage[2], "msm_g2_s": stage[3]}


def bench_rep3_party2(curve, log_m, threads, threads_b, seed=1):
    """the same workload timed with two thread settings on the same inputs: (seconds, stages), (seconds_b, stages_b), msm_shared —
    msm_shared: both settings cover every MSM window, the second run re-timed only the non-MSM stages (oracle/bench.hpp)"""
    lib().orc_bench_rep3_party2.restype = C.c_double
    sa, sb, tb, shared = (C.c_double * 4)(), (C.c_double * 4)(), C.c_double(0), C.c_int(0)
    t = lib().orc_bench_rep3_party2(curve, int(log_m), int(threads), int(threads_b), C.c_uint64(seed), sa, sb, C.byref(tb), C.byref(shared))
    if t < 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())
    names = ("spmv_pointwise_s", "ntt_s", "msm_g1_s", "msm_g2_s")
    return (t, dict(zip(names, sa))), (tb.value, dict(zip(names, sb))), bool(shared.value)


def bench_mask_draws(curve, m, seed=1):
    """seconds for the 4 x m host F::rand draws (two mul_vec calls, rngs.rs:37-46) of one proof, one thread"""
    lib().orc_bench_mask_draws.restype = C.c_double
    t = lib().orc_bench_mask_draws(curve, C.c_size_t(int(m)), C.c_uint64(seed))
    if t < 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())
    return t


def bench_rep3_party_file(curve, zkey_path, wtns_path, threads, reps=1):
    """seconds per proof for ONE REP3 party on a zkey + wtns pair, host mask draws included (+ per-stage seconds)"""
    lib().orc_bench_rep3_party_file.restype = C.c_double
    stage = (C.c_double * 5)()
    t = lib().orc_bench_rep3_party_file(curve, zkey_path.encode(), wtns_path.encode(), int(threads), int(reps), stage)
    if t < 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())
    return t, dict(zip(("rows_products_s", "ntt_s", "msm_g1_s", "msm_g2_tail_s", "mask_draws_s"), stage))


def make_synthetic(curve, log_m, seed, zkey_path, wtns_path, threads=8, n_public=1):
    """synthetic satisfiable R1CS (m - n_public - 1 constraints, n_public public inputs) with a valid Groth16 CRS, as .zkey + .wtns files"""
    if n_public == 1:
        _chk(lib().orc_make_synthetic(curve, int(log_m), C.c_uint64(seed), zkey_path.encode(), wtns_path.encode(), int(threads)))
    else:
        _chk(lib().orc_make_synthetic_pub(curve, int(log_m), C.c_uint64(seed), zkey_path.encode(), wtns_path.encode(), int(threads), C.c_uint64(int(n_public))))


# ---- snarkjs JSON <-> packed arrays ---------------------------------------------------------------
def g1_from_json(curve, arr):
    """["x","y","1"] / ["0","1","0"] (traits.rs:186-233)"""
    if arr[2] == "0":
        return np.zeros(2 * nlimbs(curve, FQ), dtype=np.uint64)
    return np.concatenate([from_dec(curve, FQ, arr[0]), from_dec(curve, FQ, arr[1])])


def g2_from_json(curve, arr):
    if arr[2][0] == "0" and arr[2][1] == "0":
        return np.zeros(4 * nlimbs(curve, FQ), dtype=np.uint64)
    return np.concatenate([from_dec(curve, FQ, arr[0][0]), from_dec(curve, FQ, arr[0][1]), from_dec(curve, FQ, arr[1][0]), from_dec(curve, FQ, arr[1][1])])


def g1_to_json(curve, pt):
    nq = nlimbs(curve, FQ)
    if not pt.any():
        return ["0", "1", "0"]
    return [to_dec(curve, FQ, pt[:nq]), to_dec(curve, FQ, pt[nq:2 * nq]), "1"]


def g2_to_json(curve, pt):
    nq = nlimbs(curve, FQ)
    d = lambda i: to_dec(curve, FQ, pt[i * nq:(i + 1) * nq])
    return [[d(0), d(1)], [d(2), d(3)], ["1", "0"]]


def proof_from_json(curve, path_or_obj):
    o = json.load(open(path_or_obj)) if isinstance(path_or_obj, str) else path_or_obj
    return np.concatenate([g1_from_json(curve, o["pi_a"]), g2_from_json(curve, o["pi_b"]), g1_from_json(curve, o["pi_c"])])


def proof_to_json(curve, proof):
    nq = nlimbs(curve, FQ)
    return {"pi_a": g1_to_json(curve, proof[:2 * nq]), "pi_b": g2_to_json(curve, proof[2 * nq:6 * nq]), "pi_c": g1_to_json(curve, proof[6 * nq:8 * nq]),
            "protocol": "groth16", "curve": "bn128" if curve == BN254 else "bls12381"}


def vk_from_json(curve, path):
    o = json.load(open(path))
    return {"alpha1": g1_from_json(curve, o["vk_alpha_1"]), "beta2": g2_from_json(curve, o["vk_beta_2"]), "gamma2": g2_from_json(curve, o["vk_gamma_2"]),
            "delta2": g2_from_json(curve, o["vk_delta_2"]), "ic": np.stack([g1_from_json(curve, p) for p in o["IC"]]), "n_public": o["nPublic"]}


def public_from_json(curve, path):
    o = json.load(open(path))
    return np.stack([from_dec(curve, FR, s) for s in o]) if o else np.zeros((0, 4), dtype=np.uint64)


# ---- seeded randomness helpers (numpy) --------------------------------------------------------------
MODULI = {
    (BN254, FR): 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    (BN254, FQ): 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    (BLS12_381, FR): 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    (BLS12_381, FQ): 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
}


def int_to_limbs(x, n):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def limbs_to_int(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def random_field(curve, which, n, rng):
    """n uniformly random reduced residues, used directly as Montgomery representatives (uniform either way)."""
    N = nlimbs(curve, which)
    p = MODULI[(curve, which)]
    pl = int_to_limbs(p, N)
    top_bits = p.bit_length() - 64 * (N - 1)
    out = np.zeros((n, N), dtype=np.uint64)
    todo = np.arange(n)
    while todo.size:
        x = rng.integers(0, 2**64, size=(todo.size, N), dtype=np.uint64)
        x[:, N - 1] &= np.uint64((1 << top_bits) - 1)
        ge = np.zeros(todo.size, dtype=bool); eq = np.ones(todo.size, dtype=bool)
        for l in range(N - 1, -1, -1):
            ge |= eq & (x[:, l] > pl[l]); eq &= x[:, l] == pl[l]
        ge |= eq
        out[todo[~ge]] = x[~ge]
        todo = todo[ge]
    return out


# ---- co-plonk round 1 (oracle/plonk.hpp) -----------------------------------------------------------------
def plonk_zkey_info(curve, path):
    info = (C.c_size_t * 6)()
    _chk(lib().orc_plonk_zkey_info(curve, path.encode(), info))
    return dict(zip(("n_vars", "n_public", "domain_size", "power", "n_additions", "n_constraints"), [int(x) for x in info]))


def plonk_zkey_data(curve, path):
    i = plonk_zkey_info(curve, path)
    nq = nlimbs(curve, FQ)
    maps = np.zeros((3, i["n_constraints"]), dtype=np.uint32)
    add_ids = np.zeros((i["n_additions"], 2), dtype=np.uint32)
    add_f = np.zeros((i["n_additions"], 2, 4), dtype=np.uint64)
    p_tau = np.zeros((i["domain_size"] + 6, 2 * nq), dtype=np.uint64)
    _chk(lib().orc_plonk_zkey_data(curve, path.encode(), _p(maps), _p(add_ids), _p(add_f), _p(p_tau)))
    return maps, add_ids, add_f, p_tau


def plonk_round1_plain(curve, path, full_witness, blind, want_polys=False):
    i = plonk_zkey_info(curve, path)
    nq = nlimbs(curve, FQ)
    out = np.zeros((3, 2 * nq), dtype=np.uint64)
    polys = np.zeros((3, i["domain_size"] + 2, 4), dtype=np.uint64) if want_polys else None
    _chk(lib().orc_plonk_round1_plain(curve, path.encode(), _p(np.ascontiguousarray(full_witness, dtype=np.uint64)),
                                      _p(np.ascontiguousarray(blind, dtype=np.uint64)), _p(out), _p(polys) if want_polys else None))
    return (out, polys) if want_polys else out


def plonk_transcript(curve, items):
    """items: list of ("scalar", limbs) / ("point", packed G1 limbs; zeros = infinity) -> Keccak256 transcript challenge (Fr)"""
    n = len(items)
    kinds = (C.c_int * n)(*[0 if k == "scalar" else 1 for k, _ in items])
    keep = [np.ascontiguousarray(v, dtype=np.uint64) for _, v in items]
    ptrs = (C.c_void_p * n)(*[v.ctypes.data for v in keep])
    out = np.zeros(4, dtype=np.uint64)
    _chk(lib().orc_plonk_transcript(curve, kinds, ptrs, n, _p(out)))
    return out


def plonk_round2_plain(curve, path, full_witness, blind, want_poly=False):
    """rounds 1 + 2 (plain driver): (beta, gamma, commit_z[, poly_z])"""
    i = plonk_zkey_info(curve, path)
    nq = nlimbs(curve, FQ)
    bg = np.zeros((2, 4), dtype=np.uint64); cz = np.zeros(2 * nq, dtype=np.uint64)
    poly = np.zeros((i["domain_size"] + 3, 4), dtype=np.uint64) if want_poly else None
    _chk(lib().orc_plonk_round2_plain(curve, path.encode(), _p(np.ascontiguousarray(full_witness, dtype=np.uint64)),
                                      _p(np.ascontiguousarray(blind, dtype=np.uint64)), _p(bg), _p(cz), _p(poly) if want_poly else None))
    return (bg[0], bg[1], cz, poly) if want_poly else (bg[0], bg[1], cz)


PLONK_COMMITS = ("a", "b", "c", "z", "t1", "t2", "t3", "wxi", "wxiw")
PLONK_CHALLENGES = ("beta", "gamma", "alpha", "xi", "v")
PLONK_EVALS = ("eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw")


def plonk_prove_plain(curve, path, full_witness, blind, upto=5, want_t=False):
    """oracle plain-driver co-plonk prover up to round `upto`: dict of commitments (PLONK_COMMITS), challenges, evaluations[, t polys]"""
    i = plonk_zkey_info(curve, path)
    nq = nlimbs(curve, FQ); n = i["domain_size"]
    commits = np.zeros((9, 2 * nq), dtype=np.uint64); ch = np.zeros((5, 4), dtype=np.uint64); ev = np.zeros((6, 4), dtype=np.uint64)
    tp = np.zeros((3 * n + 8, 4), dtype=np.uint64) if want_t else None
    _chk(lib().orc_plonk_prove_plain(curve, path.encode(), _p(np.ascontiguousarray(full_witness, dtype=np.uint64)), _p(np.ascontiguousarray(blind, dtype=np.uint64)),
                                     int(upto), _p(commits), _p(ch), _p(ev), _p(tp) if want_t else None))
    out = dict(zip(PLONK_COMMITS, commits)); out.update(zip(PLONK_CHALLENGES, ch)); out.update(zip(PLONK_EVALS, ev))
    if want_t: out.update(t1_poly=tp[:n + 1], t2_poly=tp[n + 1:2 * n + 2], t3_poly=tp[2 * n + 2:])
    return out


def plonk_verify(curve, zkey_path, proof, pub):
    """oracle Plonk verifier (co-plonk/src/plonk.rs:133-271), verifying key from the zkey header.  proof: dict with PLONK_COMMITS and PLONK_EVALS"""
    commits = np.ascontiguousarray(np.stack([proof[k] for k in PLONK_COMMITS]), dtype=np.uint64)
    evals = np.ascontiguousarray(np.stack([proof[k] for k in PLONK_EVALS]), dtype=np.uint64)
    pub = np.ascontiguousarray(pub, dtype=np.uint64).reshape(-1, 4)
    return bool(_chk(lib().orc_plonk_verify(curve, zkey_path.encode(), _p(commits), _p(evals), _p(pub), C.c_size_t(pub.shape[0]), None)))


def plonk_zkey_vk(curve, zkey_path):
    nq = nlimbs(curve, FQ)
    out = np.zeros(20 * nq + 8, dtype=np.uint64)
    _chk(lib().orc_plonk_verify(curve, zkey_path.encode(), None, None, None, C.c_size_t(0), _p(out)))
    g1 = out[:16 * nq].reshape(8, 2 * nq)
    return dict(zip(("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"), g1), X_2=out[16 * nq:20 * nq], k1=out[20 * nq:20 * nq + 4], k2=out[20 * nq + 4:])


def plonk_proof_from_json(curve, path):
    o = json.load(open(path))
    d = {k: g1_from_json(curve, o[j]) for k, j in zip(PLONK_COMMITS, ("A", "B", "C", "Z", "T1", "T2", "T3", "Wxi", "Wxiw"))}
    d.update({k: from_dec(curve, FR, o[k]) for k in PLONK_EVALS})
    return d


# ---- randomness streams (oracle/rngs.hpp) ----------------------------------------------------------------------------
def chacha_block(rounds, key_words, counter, stream=0):
    key = np.ascontiguousarray(key_words, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    lib().orc_chacha_block(rounds, _p(key), C.c_uint64(counter), C.c_uint64(stream), _p(out))
    return out


def chacha12_fr_rand(curve, seed, word_pos, n):
    """n x Fr::rand over ChaCha12Rng::from_seed(seed) at word_pos -> (n x 4 limbs, word position afterwards)"""
    seed = np.frombuffer(bytes(seed), dtype=np.uint8).copy()
    out = np.zeros((n, 4), dtype=np.uint64)
    after = C.c_uint64(0)
    _chk(lib().orc_chacha12_fr_rand(curve, _p(seed), C.c_uint64(word_pos), C.c_size_t(n), _p(out), C.byref(after)))
    return out, after.value


def rep3_masks_chacha12(curve, seed1, pos1, seed2, pos2, n):
    """n x Rep3Rand::masking_field_element -> (masks, pos1 after, pos2 after)"""
    s1 = np.frombuffer(bytes(seed1), dtype=np.uint8).copy(); s2 = np.frombuffer(bytes(seed2), dtype=np.uint8).copy()
    out = np.zeros((n, 4), dtype=np.uint64)
    p1, p2 = C.c_uint64(pos1), C.c_uint64(pos2)
    _chk(lib().orc_rep3_masks_chacha12(curve, _p(s1), C.byref(p1), _p(s2), C.byref(p2), C.c_size_t(n), _p(out)))
    return out, p1.value, p2.value
