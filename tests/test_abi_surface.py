"""CPU-only checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol the header declares,
refuses to run without a GPU (no CPU fallback), and its O(1) HOST helpers (field / point algebra compiled from the same
sources as the device code) agree with the oracle."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as orc
from oracle_lib import BN254, BLS12_381, FR, FQ, G1, G2
from product import cg, ensure_built, ROOT


def test_header_symbols_exported():
    ensure_built()
    hdr = open(os.path.join(ROOT, "include", "cogroth16_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(cg_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    lib = ctypes.CDLL(cg.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"symbols declared in the header but not exported: {missing}"
    assert sorted(cg.ABI_SYMBOLS) == declared
    assert b"gfx950" in cg.load().cg_version()


def test_rust_ffi_declares_every_header_function():
    """rust/mpc-core-hip/src/ffi.rs is generated from include/cogroth16_hip.h (scripts/gen_rust_ffi.py): the committed file equals a fresh
    generation, and the functions it declares are exactly the ones the header does — the Rust binding cannot lag the C ABI"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "scripts", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    committed = open(os.path.join(ROOT, "rust", "mpc-core-hip", "src", "ffi.rs")).read()
    assert committed == gen.generate(), "rust/mpc-core-hip/src/ffi.rs is stale: run python scripts/gen_rust_ffi.py"
    hdr = open(os.path.join(ROOT, "include", "cogroth16_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(cg_[a-z0-9_]+)\s*\(", hdr)))
    in_rust = sorted(set(re.findall(r"pub fn (cg_[a-z0-9_]+)\(", committed)))
    assert in_rust == declared
    # the option ids and the statistics record travel by value: same numbers, same field order
    for k, v in re.findall(r"(CG_OPT_[A-Z0-9_]+) = (\d+)", hdr):
        if not k.endswith("_"):
            assert f"pub const {k}: i32 = {v};" in committed
    # the other Rust sources only call functions the binding declares
    for f in ("gpu.rs", "rep3.rs", "plain.rs", "shamir.rs", "session.rs"):
        src = open(os.path.join(ROOT, "rust", "mpc-core-hip", "src", f)).read()
        used = set(re.findall(r"\b(cg_[a-z0-9_]+)\s*\(", src))
        assert used <= set(declared) | {"cg_last_error"}, (f, sorted(used - set(declared)))


def test_host_mirror_header_matches_the_library():
    """include/cogroth16_host.h declares exactly the cgh_* entry points libcogroth16_host.so exports (the host mirror is compiled against
    the header, so the signatures agree as well)"""
    import subprocess
    ensure_built()
    hdr = open(os.path.join(ROOT, "include", "cogroth16_host.h")).read()
    declared = sorted(set(re.findall(r"\b(cgh_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    lib = ctypes.CDLL(cg.HOST_LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"declared in cogroth16_host.h but not exported: {missing}"
    nm = subprocess.run(["nm", "-D", "--defined-only", cg.HOST_LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\bT (cgh_[a-z0-9_]+)$", nm, flags=re.M)))
    assert exported == declared, f"exported but not declared: {sorted(set(exported) - set(declared))}"


def test_environment_variables_are_documented():
    """The release libraries read at most 10 environment variables (VERDICT r5 #7c), every one of them in the table of its header
    (include/*.h): resources and diagnostics only.  The A/B knobs of the measurement scripts go through tune_env(), which is getenv only in
    -DCG_DEBUG_KNOBS builds; their names are listed in the headers as well."""
    import glob, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    release = set()
    for header, pattern in (("cogroth16_hip.h", "collaborative-circom_amd/csrc/*"), ("cogroth16_host.h", "collaborative-circom_amd/host/*")):
        text = open(os.path.join(root, "include", header)).read()
        env, knobs = set(), set()
        for f in glob.glob(os.path.join(root, pattern)):
            if os.path.isfile(f) and f.endswith((".hip", ".hpp", ".cpp")):
                src = open(f).read()
                # getenv inside an `#ifdef CG_DEBUG_KNOBS` block belongs to the planning build
                planning = "".join(re.findall(r"#ifdef CG_DEBUG_KNOBS(.*?)#e(?:lse|ndif)", src, flags=re.S))
                knobs |= set(re.findall(r'tune_env\("([A-Z0-9_]+)"\)', src)) | set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', planning))
                env |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', src)) - set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', planning))
        missing = sorted(n for n in env | knobs if n not in text)
        assert not missing, f"{header} does not document {missing}"
        release |= env
    assert release <= {"CG_DEV_CACHE_MB", "CG_HOST_CACHE_MB", "CG_DEBUG_ALLOC", "CG_DEBUG_STREAMS", "CGH_TIMING", "CGH_SKIP_ZKEY_VALIDATION"}, sorted(release)
    assert len(release) <= 10


def test_release_libraries_hold_no_knob_that_changes_results():
    """CG_DEBUG_NO_REDUCE (skips the bucket reductions) and CGH_EMULATE_DEVICE / CGH_EMULATE_PRIMARY_ONLY (turn devices of a multi-device
    proof into no-ops) exist only in -DCG_DEBUG_KNOBS builds: the shipped binaries do not even contain the names, so no
    environment can make them return wrong results."""
    ensure_built()
    for path in (cg.LIB_PATH, os.path.join(os.path.dirname(cg.LIB_PATH), "libcogroth16_host.so")):
        blob = open(path, "rb").read()
        for name in (b"CG_DEBUG_NO_REDUCE", b"CGH_EMULATE_DEVICE", b"CGH_EMULATE_PRIMARY_ONLY"):
            assert name not in blob, f"{os.path.basename(path)} contains {name.decode()}"


def test_no_cpu_fallback_without_gpu():
    import torch
    ensure_built()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(cg.BackendError) as e:
        cg.Context(0)
    assert "no HIP device" in str(e.value) or "error" in str(e.value)


def jac_of(curve, group, affine):
    return cg.point_from_affine(curve, group, affine)


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
@pytest.mark.parametrize("group", [G1, G2])
def test_host_point_helpers_match_oracle(curve, group):
    ensure_built()
    rng = np.random.default_rng(21 + curve * 2 + group)
    ks = orc.random_field(curve, FR, 4, rng)
    P = orc.generator_mul(curve, group, ks[0]); Q = orc.generator_mul(curve, group, ks[1])
    jp, jq = jac_of(curve, group, P), jac_of(curve, group, Q)
    np.testing.assert_array_equal(cg.point_to_affine(curve, group, jp), P)
    # add, double (P+P), P + (-P), P + inf
    np.testing.assert_array_equal(cg.point_to_affine(curve, group, cg.point_add(curve, group, jp, jq)), orc.point_add(curve, group, P, Q))
    np.testing.assert_array_equal(cg.point_to_affine(curve, group, cg.point_add(curve, group, jp, jp)), orc.point_add(curve, group, P, P))
    assert not cg.point_to_affine(curve, group, cg.point_add(curve, group, jp, cg.point_neg(curve, group, jp))).any()
    inf = jac_of(curve, group, np.zeros_like(P))
    np.testing.assert_array_equal(cg.point_to_affine(curve, group, cg.point_add(curve, group, inf, jp)), P)
    # the product's Jacobian output is accepted by the oracle's normaliser too
    np.testing.assert_array_equal(orc.jacobian_to_affine(curve, group, cg.point_add(curve, group, jp, jq)), orc.point_add(curve, group, P, Q))
    # scalar mul
    got = cg.point_to_affine(curve, group, cg.point_scalar_mul(curve, group, jp, ks[2]))
    np.testing.assert_array_equal(got, orc.points_mul(curve, group, P[None, :], ks[2][None, :])[0])
    assert not cg.point_to_affine(curve, group, cg.point_scalar_mul(curve, group, jp, np.zeros(4, dtype=np.uint64))).any()


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
@pytest.mark.parametrize("group", [G1, G2])
def test_host_scalar_mul_and_fixed_base_tables(curve, group):
    """the 64-bit-limb host arithmetic of proof assembly (csrc/host_ec64.hpp): windowed variable-base products and the 8-bit window tables
    of a session's fixed bases (cg_fixed_base_*: delta_1, delta_2, generators, public-input records) against the oracle's scalar
    multiplication, with the scalars 0, 1, r - 1, r - 3, r - 2^64 and the point at infinity"""
    ensure_built()
    rng = np.random.default_rng(71 + curve * 2 + group)
    ks = orc.random_field(curve, FR, 8, rng)
    ks[5] = 0; ks[6] = orc.from_dec(curve, FR, 1); ks[7] = orc.from_dec(curve, FR, orc.MODULI[(curve, FR)] - 1)
    ks[3] = orc.from_dec(curve, FR, orc.MODULI[(curve, FR)] - 3); ks[4] = orc.from_dec(curve, FR, orc.MODULI[(curve, FR)] - (1 << 64))   # small negative (negated product) / just not small
    P = orc.generator_mul(curve, group, ks[0])
    jp = jac_of(curve, group, P)
    want = [orc.points_mul(curve, group, P[None, :], k[None, :])[0] for k in ks]
    for k, w in zip(ks, want):
        np.testing.assert_array_equal(cg.point_to_affine(curve, group, cg.point_scalar_mul(curve, group, jp, k)), w)
    # a non-normalised Jacobian input (Z != 1): the product of a product
    jq = cg.point_scalar_mul(curve, group, jp, ks[1])
    k12 = orc.field_op(curve, FR, "mul", ks[1][None, :], ks[2][None, :])[0]
    np.testing.assert_array_equal(cg.point_to_affine(curve, group, cg.point_scalar_mul(curve, group, jq, ks[2])), orc.points_mul(curve, group, P[None, :], k12[None, :])[0])
    fb = cg.FixedBase(curve, group, jq)                                # table of a base with Z != 1
    for k in ks:
        np.testing.assert_array_equal(cg.point_to_affine(curve, group, fb.mul(k)), cg.point_to_affine(curve, group, cg.point_scalar_mul(curve, group, jq, k)))
    fb.close()
    fb = cg.FixedBase(curve, group, jp)
    for k, w in zip(ks, want):
        np.testing.assert_array_equal(cg.point_to_affine(curve, group, fb.mul(k)), w)
    fb.close()
    inf = jac_of(curve, group, np.zeros_like(P))
    assert not cg.point_to_affine(curve, group, cg.point_scalar_mul(curve, group, inf, ks[1])).any()
    fbi = cg.FixedBase(curve, group, inf)
    assert not cg.point_to_affine(curve, group, fbi.mul(ks[1])).any()
    fbi.close()


@pytest.mark.parametrize("curve", [BN254, BLS12_381])
def test_host_fr_ops_match_oracle(curve):
    ensure_built()
    rng = np.random.default_rng(33)
    a, b = orc.random_field(curve, FR, 2, rng)
    for op in ("add", "sub", "mul"):
        np.testing.assert_array_equal(cg.fr_op(curve, op, a, b), orc.field_op(curve, FR, op, a, b))
    np.testing.assert_array_equal(cg.fr_op(curve, "inv", a), orc.field_inverse(curve, FR, a))
    # edge values: 0, 1, p-1
    p1 = orc.from_dec(curve, FR, orc.MODULI[(curve, FR)] - 1)
    one = orc.from_dec(curve, FR, 1); zero = np.zeros(4, dtype=np.uint64)
    np.testing.assert_array_equal(cg.fr_op(curve, "add", p1, one), zero)
    np.testing.assert_array_equal(cg.fr_op(curve, "sub", zero, one), p1)
    np.testing.assert_array_equal(cg.fr_op(curve, "mul", p1, p1), one)


def test_process_wide_options_round_trip_and_refuse_unknown_ids():
    """cg_set_option / cgh_set_option (the homes of what used to be environment variables): defaults as documented in the headers, values
    round-trip, unknown options and out-of-range values are errors, not silent no-ops.  No device needed."""
    ensure_built()
    assert [cg.get_option(o) for o in (cg.GOPT_SUBGROUP_FULL, cg.GOPT_COMPACT_MIN_LOG, cg.GOPT_SORT_STAGING, cg.GOPT_SORT_SMALL, cg.GOPT_MSM_STAGED_OUT, cg.GOPT_STREAM_PROBES)] == [0, 14, 1, 1, 0, 1]
    cg.set_option(cg.GOPT_COMPACT_MIN_LOG, 64); assert cg.get_option(cg.GOPT_COMPACT_MIN_LOG) == 64
    cg.set_option(cg.GOPT_COMPACT_MIN_LOG, 14)
    for bad in ((0, 1), (99, 1), (cg.GOPT_SUBGROUP_FULL, 2), (cg.GOPT_SORT_SMALL, -1), (cg.GOPT_COMPACT_MIN_LOG, 65)):
        with pytest.raises(cg.BackendError):
            cg.set_option(*bad)
    defaults = {cg.HOST_OPT_XCHG_ASYNC_MIN: 1 << 17, cg.HOST_OPT_DEVICE_MASKS_MIN: 1 << 11, cg.HOST_OPT_XCHG_COPY_STREAM_MIN: 1 << 14, cg.HOST_OPT_SECOND_CONTEXT_MIN_LOG: 15,
                cg.HOST_OPT_DISTRIBUTED_MAP: 1, cg.HOST_OPT_ONE_CONTEXT: 0, cg.HOST_OPT_SPLIT_FIRST_MSM_MIN: 0, cg.HOST_OPT_CTX_WIDE_LOG: 0, cg.HOST_OPT_CTX_OFF_MAIN_LOG: 0,
                cg.HOST_OPT_CTX_SOLO_LOG: 0}
    for opt, want in defaults.items():
        assert cg.host_get_option(opt) == want, opt
    with cg.host_options({cg.HOST_OPT_XCHG_ASYNC_MIN: 4096}):
        assert cg.host_get_option(cg.HOST_OPT_XCHG_ASYNC_MIN) == 4096
    assert cg.host_get_option(cg.HOST_OPT_XCHG_ASYNC_MIN) == 1 << 17
    for bad in ((0, 1), (len(defaults) + 1, 1), (cg.HOST_OPT_ONE_CONTEXT, -1)):
        with pytest.raises(cg.BackendError):
            cg.host_set_option(*bad)


def test_planning_build_of_the_host_library_holds_the_knobs():
    """`make -C collaborative-circom_amd/host KNOBS=1` (what scripts/multi_device_emulation.py needs) still builds, exports the same entry points and is
    the one place where CGH_EMULATE_DEVICE and the A/B knobs exist"""
    ensure_built()
    host_dir = os.path.join(ROOT, "collaborative-circom_amd", "host")
    subprocess.run(["make", "-C", host_dir, "KNOBS=1", "-j4"], check=True, capture_output=True)
    knobs = os.path.join(ROOT, "collaborative-circom_amd", "libcogroth16_host_knobs.so")
    blob = open(knobs, "rb").read()
    for name in (b"CGH_EMULATE_DEVICE", b"CGH_BULK_CHUNK", b"CGH_G2_AFTER"):
        assert name in blob, name
    sym = lambda path: sorted(set(re.findall(r"\bT (cgh_[a-z0-9_]+)$", subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout, flags=re.M)))
    assert sym(knobs) == sym(cg.HOST_LIB_PATH if not os.environ.get("COGROTH16_HOST_LIB") else os.path.join(ROOT, "collaborative-circom_amd", "libcogroth16_host.so"))
