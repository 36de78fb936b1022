"""Context classes (cg_ctx_create_ex: chain / bulk), parked device and page-locked blocks (cg_dev_alloc / cg_dev_free,
cg_host_alloc / cg_host_free), and the sliced G2 accumulation of a context that runs next to a latency chain (cg_msm_set_chunk).
None of this may change a result: every check is bit-exact against the oracle or against the default context."""
import numpy as np
import pytest

import oracle_lib as orc
from oracle_lib import BN254, FR, G1, G2
from product import cg, ensure_built

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def built():
    ensure_built()


def test_chain_and_bulk_contexts_compute_the_same(built):
    """the three context classes differ in stream priorities and hardware queues only"""
    rng = np.random.default_rng(77)
    n = 5000
    a, b = orc.random_field(BN254, FR, n, rng), orc.random_field(BN254, FR, n, rng)
    pts = np.stack([orc.generator_mul(BN254, G1, s) for s in orc.random_field(BN254, FR, 32, rng)])[rng.integers(0, 32, size=n)]
    want_mul = orc.field_op(BN254, FR, "mul", a, b)
    want_msm = orc.msm(BN254, G1, pts, a)
    for flags in (0, cg.Context.CHAIN, cg.Context.BULK):
        c = cg.Context(0, flags)
        da, db = c.to_device(a), c.to_device(b)
        out = c.alloc(n * 32)
        c.vec_mul(BN254, out, da, db, n)
        np.testing.assert_array_equal(out.download((n, 4)), want_mul)
        # asynchronous copies: the chain context has its copy streams from the start, the others create them here
        h = c.host_alloc((n, 4)); h[:] = b
        t = c.upload_begin(out, h, after_stream=True); c.copy_wait(t)
        np.testing.assert_array_equal(out.download((n, 4)), b)
        c.host_free(h)
        bases = c.register_bases(BN254, G1, pts)
        got = c.msm(bases, [a])
        np.testing.assert_array_equal(cg.point_to_affine(BN254, G1, got[0]), want_msm)
        bases.release(); c.close()


def test_contexts_of_a_stream_group_compute_the_same(built):
    """cg_stream_group_begin / _end place the streams of one party's contexts on hardware queues of their own: placement only.  Groups nest,
    an unmatched _end is harmless, contexts made in a group work after it has ended and after other contexts have come and gone"""
    rng = np.random.default_rng(78)
    n = 4096
    a, b = orc.random_field(BN254, FR, n, rng), orc.random_field(BN254, FR, n, rng)
    want = orc.field_op(BN254, FR, "mul", a, b)
    lib = cg.load()
    cg._chk(lib.cg_stream_group_end())                              # nothing open: no effect
    cg._chk(lib.cg_stream_group_begin()); cg._chk(lib.cg_stream_group_begin())
    chain, bulk = cg.Context(0, cg.Context.CHAIN), cg.Context(0, cg.Context.BULK)
    cg._chk(lib.cg_stream_group_end())
    third = cg.Context(0)                                           # still inside the outer group
    cg._chk(lib.cg_stream_group_end())
    outside = cg.Context(0)
    outside.close()
    for c in (chain, bulk, third):
        da, db = c.to_device(a), c.to_device(b)
        out = c.alloc(n * 32)
        c.vec_mul(BN254, out, da, db, n)
        np.testing.assert_array_equal(out.download((n, 4)), want)
        for x in (da, db, out): x.free()
    for c in (chain, bulk, third): c.close()
    # the parked streams serve the next group as well
    cg._chk(lib.cg_stream_group_begin())
    again = [cg.Context(0, f) for f in (cg.Context.CHAIN, cg.Context.BULK, 0, 0)]
    cg._chk(lib.cg_stream_group_end())
    for c in again:
        da, db = c.to_device(a), c.to_device(b); out = c.alloc(n * 32)
        c.vec_mul(BN254, out, da, db, n)
        np.testing.assert_array_equal(out.download((n, 4)), want)
        c.close()


def test_released_blocks_are_handed_out_again_and_hold_no_stale_work(built):
    """cg_dev_free parks a block behind the work enqueued so far; the next allocation of that size gets it back only once that work is done"""
    c = cg.Context(0)
    rng = np.random.default_rng(5)
    n = 1 << 16
    a, b = orc.random_field(BN254, FR, n, rng), orc.random_field(BN254, FR, n, rng)
    da, db = c.to_device(a), c.to_device(b)
    want = orc.field_op(BN254, FR, "mul", a, b)
    seen = set()
    for rep in range(6):
        out = c.alloc(n * 32)
        seen.add(out.ptr)
        for _ in range(20): c.vec_mul(BN254, out, da, db, n)       # the stream is still busy with these when the block is released
        got_before_free = out.download((n, 4))
        np.testing.assert_array_equal(got_before_free, want)
        c.vec_mul(BN254, out, db, da, n)                            # enqueued, not waited for
        out.free()
        other = c.alloc(n * 32); other.zero()                       # same size: may be the parked block — never while the product above is pending
        np.testing.assert_array_equal(other.download((n, 4)), np.zeros((n, 4), dtype=np.uint64))
        other.free()
    # (how many distinct blocks the loop above saw depends on how soon the events behind the parked blocks complete: 3 to 6 on the boxes seen)
    # reuse itself, deterministically: a block released on idle streams comes back for the next request of its size once its event has fired
    import time
    blk = c.alloc(n * 32); blk.zero(); c.sync()
    parked = blk.ptr; blk.free()
    got_back = False
    for _ in range(200):
        nxt = c.alloc(n * 32); hit = nxt.ptr == parked or nxt.ptr in seen
        nxt.free()
        if hit: got_back = True; break
        time.sleep(0.002)
    assert got_back, "released blocks were not reused"
    # a size nobody released: a fresh block; zero-size and odd sizes round up
    for nbytes in (1, 17, 4097, (1 << 16) + 8):
        blk = c.alloc(nbytes); blk.zero(); blk.free()
    # cg_dev_cache_trim: the parked blocks go back to the runtime (at least the ones released above), live blocks are untouched
    keep = c.to_device(a)
    big = c.alloc(3 << 20); big.zero(); c.sync(); big.free()
    released = cg.dev_cache_trim(0)
    assert released >= (3 << 20), released
    assert cg.dev_cache_trim(0) == 0                                # nothing parked any more
    np.testing.assert_array_equal(keep.download((n, 4)), a)         # a live block survives the trim
    again = c.alloc(3 << 20); again.zero(); again.free(); keep.free()
    with pytest.raises(cg.BackendError):
        cg.dev_cache_trim(1 << 20)                                  # no such device
    c.close()


def test_blocks_released_together_share_one_mark(built):
    """cg_dev_free_many: n blocks parked behind one release mark — none of them is handed out while the work enqueued before the release is
    pending, all of them come back afterwards, NULL entries are skipped, the cache accounting (cg_dev_cache_trim) sees every block"""
    import time
    c = cg.Context(0)
    cg.dev_cache_trim(0)
    rng = np.random.default_rng(6)
    n = 1 << 15
    a, b = orc.random_field(BN254, FR, n, rng), orc.random_field(BN254, FR, n, rng)
    da, db = c.to_device(a), c.to_device(b)
    want = orc.field_op(BN254, FR, "mul", a, b)
    sizes = [n * 32, n * 32 + 4096, n * 32 + 8192, n * 32 + 12288]
    outs = [c.alloc(sz) for sz in sizes]
    ptrs = {o.ptr for o in outs}
    for o in outs:
        for _ in range(10): c.vec_mul(BN254, o, da, db, n)
    np.testing.assert_array_equal(outs[3].download((n, 4)), want)
    for o in outs: c.vec_mul(BN254, o, db, da, n)                   # enqueued, not waited for
    c.free_many(outs)
    assert all(o.ptr == 0 for o in outs)
    # same sizes again: zeroed and read back — a block handed out too early would still be written by the products above
    again = [c.alloc(sz) for sz in sizes]
    for o in again: o.zero()
    for o in again: np.testing.assert_array_equal(o.download((n, 4)), np.zeros((n, 4), dtype=np.uint64))
    c.free_many(again); c.sync()
    got_back = set()
    for _ in range(200):
        trial = [c.alloc(sz) for sz in sizes]
        got_back |= {t.ptr for t in trial} & (ptrs | {o.ptr for o in again})
        c.free_many(trial)
        if len(got_back) >= 2: break
        time.sleep(0.002)
    assert len(got_back) >= 2, "blocks released together were not reused"
    c.sync()
    assert cg.dev_cache_trim(0) >= sum(sizes)                       # every block of the batch was parked
    # an empty batch and a batch of NULLs are fine
    import ctypes as C
    assert cg.load().cg_dev_free_many(c.h, (C.c_void_p * 2)(None, None), C.c_size_t(2)) == 0
    assert cg.load().cg_dev_free_many(c.h, None, C.c_size_t(0)) == 0
    da.free(); db.free(); c.close()


def test_pinned_blocks_are_parked(built):
    c = cg.Context(0)
    h1 = c.host_alloc((1 << 18, 4)); h1[:] = 7
    p1 = h1.ctypes.data
    assert cg.load().cg_host_is_pinned(cg.C.c_void_p(p1)) == 1
    c.host_free(h1)
    h2 = c.host_alloc((1 << 18, 4))
    assert h2.ctypes.data == p1, "a released page-locked block of the same size should be handed out again"
    assert cg.load().cg_host_is_pinned(cg.C.c_void_p(h2.ctypes.data)) == 1
    d = c.alloc(h2.nbytes)
    h2[:] = 9
    c.copy_wait(c.upload_begin(d, h2, after_stream=False))
    assert (d.download((1 << 18, 4)) == 9).all()
    c.host_free(h2); c.close()


@pytest.mark.parametrize("group", [G1, G2])
def test_msm_next_to_a_chain_equals_the_default(built, group):
    """a bulk context with a chunk request (short-lived workgroups; G2: the accumulation launched one chip-load at a time, several
    launches at this size) returns the same points as the default context"""
    log_n = 20
    n = 1 << log_n
    rng = np.random.default_rng(41)
    base_scalars, sc = orc.random_field(BN254, FR, n, rng), orc.random_field(BN254, FR, n, rng)
    outs = []
    for flags, chunk in ((0, 0), (cg.Context.BULK, 64), (cg.Context.BULK, 16)):
        c = cg.Context(0, flags)
        if chunk: c.msm_set_chunk(chunk)
        d_base, d_sc = c.to_device(base_scalars), c.to_device(sc)
        bases = c.bases_from_scalars(BN254, group, d_base, n)
        c.precompute_bases(bases, 0)
        out = c.msm_end(c.msm_dev_begin_multi([bases], [d_sc], n)[0])
        outs.append(cg.point_to_affine(BN254, group, out[0]))
        bases.release(); c.close()
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], outs[2])
    # and the value itself: sum_i s_i * (b_i * G) = (sum_i s_i b_i) * G
    acc = orc.field_op(BN254, FR, "mul", base_scalars, sc)
    while acc.shape[0] > 1:
        acc = orc.field_op(BN254, FR, "add", acc[: acc.shape[0] // 2], acc[acc.shape[0] // 2:])
    np.testing.assert_array_equal(outs[0], orc.generator_mul(BN254, group, acc[0]))


@pytest.mark.parametrize("log_n,window", [(10, 8), (14, 0), (17, 0)])
def test_transforms_run_beside_the_reductions_of_an_msm_in_flight(built, log_n, window):
    """one context: an MSM is begun (schedule, accumulation and reductions enqueued on the side streams), transforms of the same length follow
    at once on the main stream, then the MSM is collected.  The transforms' scratch is a block of its own — they no longer wait for the bucket
    reductions that read the MSM arena — so both must still be exact: the transform against the oracle, the MSM against its closed form.
    Sizes on both sides of the off-main-stream accumulation bound; twice, so that the second MSM finds the first one's scratch slots busy."""
    n = 1 << log_n
    rng = np.random.default_rng(1234 + log_n)
    _, roots, _ = orc.roots_of_unity(BN254)
    c = cg.Context(0)
    base_scalars, sc = orc.random_field(BN254, FR, n, rng), orc.random_field(BN254, FR, n, rng)
    d_base, d_sc = c.to_device(base_scalars), c.to_device(sc)
    tables = []
    for group in (G1, G2):
        bases = c.bases_from_scalars(BN254, group, d_base, n)
        c.precompute_bases(bases, window)
        tables.append(bases)
    x = orc.random_field(BN254, FR, n, rng)
    want_x = orc.ntt(BN254, x, roots[log_n])
    for _ in range(2):
        dx = c.to_device(x)
        tickets = c.msm_dev_begin_multi(tables, [d_sc], n)
        c.ntt_dev(BN254, [dx], n, roots[log_n])
        np.testing.assert_array_equal(dx.download((n, 4)), want_x)
        c.ntt_dev(BN254, [dx], n, roots[log_n], inverse=True)
        for group, t in zip((G1, G2), tickets):
            np.testing.assert_array_equal(cg.point_to_affine(BN254, group, c.msm_end(t)[0]), _closed_form(base_scalars, sc, group))
        np.testing.assert_array_equal(dx.download((n, 4)), x)
    for bases in tables: bases.release()
    c.close()


def _closed_form(base_scalars, sc, group):
    acc = orc.field_op(BN254, FR, "mul", base_scalars, sc)
    while acc.shape[0] > 1:
        acc = orc.field_op(BN254, FR, "add", acc[: acc.shape[0] // 2], acc[acc.shape[0] // 2:])
    return orc.generator_mul(BN254, group, acc[0])


def test_sort_schedule_variants_agree(built):
    """the scalar-side schedule has three shapes at this size: staged scatters (default), the record-per-lane scatters they replaced
    (kept for key spaces of more than 2 048 regions: here a classic MSM with window 18, 15 bucket sets = 3 840 regions) and the
    one-pass counting sort of small inputs; all give sum_i s_i (b_i G) = (sum_i s_i b_i) G"""
    n = 1 << 18
    rng = np.random.default_rng(4242)
    base_scalars, sc = orc.random_field(BN254, FR, n, rng), orc.random_field(BN254, FR, n, rng)
    want = _closed_form(base_scalars, sc, G1)
    c = cg.Context(0)
    d_base, d_sc = c.to_device(base_scalars), c.to_device(sc)
    bases = c.bases_from_scalars(BN254, G1, d_base, n)
    for window in (0, 18, 16, 13):                          # automatic, 3 840 regions (legacy scatters), 1 024 regions (staged), 80 regions
        c.set_msm_window(window)
        out = c.msm_end(c.msm_dev_begin_multi([bases], [d_sc], n)[0])
        np.testing.assert_array_equal(cg.point_to_affine(BN254, G1, out[0]), want, err_msg=f"window {window}")
    c.set_msm_window(0)
    c.precompute_bases(bases, 0)                            # shared bucket set
    out = c.msm_end(c.msm_dev_begin_multi([bases], [d_sc], n)[0])
    np.testing.assert_array_equal(cg.point_to_affine(BN254, G1, out[0]), want)
    bases.release(); c.close()


def test_legacy_scatters_and_release_at_once_in_a_fresh_process(built):
    """CG_GOPT_SORT_STAGING = 0 (the former scatter kernels on the default shape; cg_set_option) and CG_DEV_CACHE_MB=0 / CG_HOST_CACHE_MB=0
    (blocks released at once; read when the library is loaded): a child process runs an MSM and a REP3 product against the closed forms under them"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
import oracle_lib as orc
from oracle_lib import BN254, FR, G1
from product import cg, ensure_built
ensure_built()
n = 1 << 18
rng = np.random.default_rng(99)
b, s = orc.random_field(BN254, FR, n, rng), orc.random_field(BN254, FR, n, rng)
acc = orc.field_op(BN254, FR, "mul", b, s)
while acc.shape[0] > 1: acc = orc.field_op(BN254, FR, "add", acc[: acc.shape[0] // 2], acc[acc.shape[0] // 2:])
want = orc.generator_mul(BN254, G1, acc[0])
cg.set_option(cg.GOPT_SORT_STAGING, 0)
c = cg.Context(0)
for rep in range(3):
    db, ds = c.to_device(b), c.to_device(s)
    bases = c.bases_from_scalars(BN254, G1, db, n); c.precompute_bases(bases, 0)
    out = c.msm_end(c.msm_dev_begin_multi([bases], [ds], n)[0])
    assert (cg.point_to_affine(BN254, G1, out[0]) == want).all()
    h = c.host_alloc((n, 4)); h[:] = s; c.copy_wait(c.upload_begin(ds, h, after_stream=False)); c.host_free(h)
    bases.release(); db.free(); ds.free()
c.close()
print("child ok")
'''
    env = dict(os.environ, CG_DEV_CACHE_MB="0", CG_HOST_CACHE_MB="0")            # (the two resource variables the library still reads; the sort staging is an option)
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("knobs", [{"staged_out": 1}, {"off_main_log": 0}, {"solo_log": 0}, {"staged_out": 1, "off_main_log": 0, "sort_small": 0}, {"one_stream_log": 20}, {"stream_probes": 0}])
def test_small_call_options_do_not_change_results_in_a_fresh_process(built, knobs):
    """the defaults for small MSM calls — sums written straight into the ticket's page-locked buffer, accumulations off the main stream, the
    closed main-stream sequence of single-field calls, the one-workgroup schedule kernel — against the other value of their options
    (cg_set_option / cg_ctx_set_option; environment variables until round 5): a child process runs G1 and G2 MSMs of 2^6 .. 2^13 points
    with a transform enqueued beside them and checks the closed forms"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
import oracle_lib as orc
from oracle_lib import BN254, FR, G1, G2
from product import cg, ensure_built
ensure_built()
rng = np.random.default_rng(5)
_, roots, _ = orc.roots_of_unity(BN254)
knobs = KNOBS
if "staged_out" in knobs: cg.set_option(cg.GOPT_MSM_STAGED_OUT, knobs["staged_out"])
if "sort_small" in knobs: cg.set_option(cg.GOPT_SORT_SMALL, knobs["sort_small"])
if "stream_probes" in knobs: cg.set_option(cg.GOPT_STREAM_PROBES, knobs["stream_probes"])
c = cg.Context(0)
for name, opt in (("off_main_log", cg.OPT_MSM_OFF_MAIN_LOG), ("solo_log", cg.OPT_MSM_SOLO_LOG), ("one_stream_log", cg.OPT_MSM_ONE_STREAM_LOG)):
    if name in knobs:
        c.set_option(opt, knobs[name]); assert c.get_option(opt) == knobs[name]
for log_n, window in ((6, 8), (9, 10), (11, 13), (13, 13)):
    n = 1 << log_n
    b, s = orc.random_field(BN254, FR, n, rng), orc.random_field(BN254, FR, n, rng)
    acc = orc.field_op(BN254, FR, "mul", b, s)
    while acc.shape[0] > 1: acc = orc.field_op(BN254, FR, "add", acc[: acc.shape[0] // 2], acc[acc.shape[0] // 2:])
    db, ds = c.to_device(b), c.to_device(s)
    tables = []
    for group in (G1, G2):
        bases = c.bases_from_scalars(BN254, group, db, n); c.precompute_bases(bases, window); tables.append(bases)
    x = orc.random_field(BN254, FR, n, rng); dx = c.to_device(x)
    tickets = c.msm_dev_begin_multi(tables, [ds], n)
    c.ntt_dev(BN254, [dx], n, roots[log_n])
    for group, t in zip((G1, G2), tickets):
        assert (cg.point_to_affine(BN254, group, c.msm_end(t)[0]) == orc.generator_mul(BN254, group, acc[0])).all(), (log_n, group)
    assert (dx.download((n, 4)) == orc.ntt(BN254, x, roots[log_n])).all(), log_n
    for bases in tables: bases.release()
c.close()
print("child ok")
'''
    r = subprocess.run([sys.executable, "-c", code.replace("KNOBS", repr(knobs))], cwd=root, env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_device_preflight_accepts_one_gpu_and_refuses_a_repeated_one(tmp_path):
    """cg_device_preflight (VERDICT r5 #4c): the first contact of a multi-device session / `bench.py --gpus N` with a node.  On one GPU: a
    single device passes; the same GPU listed twice is refused with both entries and the PCI bus id named, and passes only under the
    explicit shared-device flag (then as a checked LOCAL copy); a device that does not exist is refused; a session over a repeated GPU does
    not open without CGH_SESSION_SHARED_DEVICES."""
    cg = ensure_built()
    rep = cg.device_preflight([0])
    assert len(rep["devices"]) == 1 and rep["devices"][0]["pci"] and rep["pairs"] == []
    with pytest.raises(cg.BackendError) as e:
        cg.device_preflight([0, 0])
    assert "SAME GPU" in str(e.value) and rep["devices"][0]["pci"] in str(e.value)
    with pytest.raises(cg.BackendError):
        cg.device_preflight([0, cg.device_count() + 3])
    rep = cg.device_preflight([0, 0, 0], allow_shared=True)
    assert len(rep["pairs"]) == 6 and all(p["same_gpu"] and p["GBs"] > 0 and p["checksum"] > 0 for p in rep["pairs"])
    if cg.device_count() > 1:                                                  # a node: the real thing
        rep = cg.device_preflight(list(range(cg.device_count())))
        assert all(p["peer_access"] and not p["same_gpu"] for p in rep["pairs"])
    zp, wp = str(tmp_path / "s.zkey"), str(tmp_path / "s.wtns")
    cg.host_synth_circuit(cg.BN254, 8, 3, zp, wp)
    with pytest.raises(cg.BackendError) as e:
        cg.ProvingSession(cg.BN254, zp, precompute=False, devices=[0, 0])
    assert "SAME GPU" in str(e.value)
    cg.ProvingSession(cg.BN254, zp, precompute=False, devices=[0, 0], shared_devices=True).close()
