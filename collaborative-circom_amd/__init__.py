"""cogroth16-hip: MI355X (gfx950) backend for the co-groth16 prover hot path of TaceoLabs/collaborative-circom.

The product is the C-ABI shared library `libcogroth16_hip.so` (include/cogroth16_hip.h) built from csrc/ with hipcc.
This module is only a thin ctypes loader over it (directory name contains a hyphen: import it with
`importlib.import_module("collaborative-circom_amd")`).  There is NO CPU fallback: `load()` raises if the library is
missing, and `Context()` raises if no HIP device is present.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libcogroth16_hip.so")
HOST_LIB_PATH = os.path.join(HERE, "libcogroth16_host.so")
# planning / A-B scripts only: another build of a library (scripts/multi_device_emulation.py: the -DCG_DEBUG_KNOBS build of the host library,
# `make -C host KNOBS=1`; scripts/ntt_timing.py: a hip library built with other -D switches)
if os.environ.get("COGROTH16_HIP_LIB"):
    LIB_PATH = os.path.abspath(os.environ["COGROTH16_HIP_LIB"])
if os.environ.get("COGROTH16_HOST_LIB"):
    HOST_LIB_PATH = os.path.abspath(os.environ["COGROTH16_HOST_LIB"])

BN254, BLS12_381 = 0, 1
G1, G2 = 0, 1

_lib = None


class BackendError(RuntimeError):
    pass


def build(bls=True, jobs=8):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc"), f"-j{jobs}", f"BLS={1 if bls else 0}"])
    if os.path.isdir(os.path.join(HERE, "host")) and os.path.exists(os.path.join(HERE, "host", "Makefile")):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "host"), f"-j{jobs}"])
    return LIB_PATH


# per-context tuning options (include/cogroth16_hip.h)
OPT_MSM_CHUNK, OPT_MSM_WINDOW, OPT_MSM_SCATTER_CAP, OPT_MSM_TABLE_ORDER, OPT_MSM_G2_SLICES, OPT_MSM_REDUCE_BATCH, OPT_MSM_ACC_SLOTS, OPT_MSM_G2_AFTER, OPT_MSM_WIDE_SMALL = 1, 2, 3, 4, 5, 6, 7, 8, 9
OPT_MSM_ONE_STREAM_LOG, OPT_MSM_OFF_MAIN_LOG, OPT_MSM_SOLO_LOG = 10, 11, 12
# process-wide options (cg_set_option)
GOPT_SUBGROUP_FULL, GOPT_COMPACT_MIN_LOG, GOPT_SORT_STAGING, GOPT_SORT_SMALL, GOPT_MSM_STAGED_OUT, GOPT_STREAM_PROBES = 1, 2, 3, 4, 5, 6

# every symbol include/cogroth16_hip.h declares (checked by tests/test_abi_surface.py)
ABI_SYMBOLS = [
    "cg_ctx_create", "cg_ctx_create_ex", "cg_ctx_destroy", "cg_ctx_sync", "cg_ctx_stream", "cg_ctx_set_stream", "cg_last_error", "cg_version",
    "cg_dev_alloc", "cg_dev_free", "cg_dev_free_many", "cg_dev_cache_trim", "cg_stream_group_begin", "cg_stream_group_end", "cg_dev_upload", "cg_dev_download", "cg_dev_memset_zero",
    "cg_bases_register", "cg_bases_register_device", "cg_bases_release", "cg_bases_len", "cg_bases_precompute", "cg_bases_check_on_curve", "cg_bases_check_subgroup",
    "cg_msm", "cg_msm_dev", "cg_msm_dev_begin", "cg_msm_dev_begin_multi", "cg_msm_end", "cg_msm_set_window", "cg_msm_set_chunk", "cg_ctx_set_option", "cg_ctx_get_option", "cg_set_option", "cg_get_option", "cg_msm_scalars_after", "cg_msm_set_scatter_capacity",
    "cg_ntt", "cg_ntt_dev", "cg_ntt_coset_pair_dev", "cg_chacha12_fr_rand_dev", "cg_chacha12_fr_rand_dev_begin", "cg_chacha12_fr_rand_dev_finish",
    "cg_host_alloc", "cg_host_free", "cg_host_is_pinned", "cg_dev_download_begin", "cg_dev_upload_begin", "cg_stream_mark", "cg_dev_download_begin_after", "cg_copy_wait", "cg_copy_fence",
    "cg_vec_add_dev", "cg_vec_sub_dev", "cg_vec_mul_dev", "cg_vec_rep3_mul_local_dev", "cg_vec_distribute_powers_dev", "cg_vec_affine_dev", "cg_vec_fill_dev", "cg_vec_gather_strided_dev", "cg_vec_lincomb_dev", "cg_vec_prefix_prod_dev", "cg_vec_prefix_sum_dev", "cg_vec_inverse_dev",
    "cg_spmv_csr_dev", "cg_vec_mul", "cg_vec_rep3_mul_local",
    "cg_point_add", "cg_point_neg", "cg_point_scalar_mul", "cg_fixed_base_create", "cg_fixed_base_mul", "cg_fixed_base_destroy", "cg_point_to_affine", "cg_point_from_affine", "cg_point_validate", "cg_fr_is_canonical", "cg_vec_check_canonical_dev", "cg_fr_op",
    "cg_fr_from_canonical", "cg_fr_to_canonical", "cg_fq_to_canonical", "cg_fq_from_canonical", "cg_point_generator",
    "cg_bases_synth_multiples", "cg_bases_download", "cg_bases_from_scalars",
    "cg_dev_copy_peer", "cg_ctx_device", "cg_device_count", "cg_device_preflight",
    "cg_stats_enable", "cg_stats",
]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError(f"{LIB_PATH} is missing: build it with `make -C collaborative-circom_amd/csrc` "
                           "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.cg_last_error.restype = C.c_char_p
    lib.cg_version.restype = C.c_char_p
    lib.cg_ctx_stream.restype = C.c_void_p
    lib.cg_ctx_stream.argtypes = [C.c_void_p]
    lib.cg_bases_len.restype = C.c_size_t
    lib.cg_bases_len.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def _chk(rc):
    if rc != 0:
        raise BackendError(f"cogroth16_hip error {rc}: {load().cg_last_error().decode()}")


def _hp(a):
    """host pointer of a numpy array (or None)"""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dp(x):
    """device pointer: DevBuf, int, torch tensor (data_ptr) or None"""
    if x is None:
        return None
    if isinstance(x, DevBuf):
        return C.c_void_p(x.ptr)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError(f"not a device pointer: {type(x)}")


def fq_limbs(curve):
    return 6 if curve == BLS12_381 else 4


def point_words(curve, group, coords):
    return fq_limbs(curve) * (1 if group == G1 else 2) * coords


class DevBuf:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = C.c_void_p()
        _chk(load().cg_dev_alloc(ctx.h, C.c_size_t(nbytes), C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _chk(load().cg_dev_upload(self.ctx.h, C.c_void_p(self.ptr), _hp(arr), C.c_size_t(arr.nbytes)))
        return self

    def download(self, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _chk(load().cg_dev_download(self.ctx.h, _hp(out), C.c_void_p(self.ptr), C.c_size_t(out.nbytes)))
        return out

    def zero(self):
        _chk(load().cg_dev_memset_zero(self.ctx.h, C.c_void_p(self.ptr), C.c_size_t(self.nbytes)))
        return self

    def free(self):
        if self.ptr:
            _chk(load().cg_dev_free(self.ctx.h, C.c_void_p(self.ptr)))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Bases:
    def __init__(self, ctx, curve, group, handle, n):
        self.ctx, self.curve, self.group, self.h, self.n = ctx, curve, group, handle, n

    def release(self):
        if self.h:
            load().cg_bases_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Context:
    """One per MPC party thread (mirrors `&mut self` of the reference drivers)."""

    CHAIN, BULK = 1, 2          # cg_ctx_create_ex flags: the context of a dependency chain / the context that fills the chip next to it

    def __init__(self, device=0, flags=0):
        h = C.c_void_p()
        _chk(load().cg_ctx_create_ex(int(device), C.c_uint32(int(flags)), C.byref(h)))
        self.h = h
        self.device = device

    def set_option(self, option, value):
        """per-context tuning table (include/cogroth16_hip.h: CG_OPT_*); never changes results"""
        _chk(load().cg_ctx_set_option(self.h, int(option), C.c_int64(int(value))))

    def get_option(self, option):
        v = C.c_int64(0)
        _chk(load().cg_ctx_get_option(self.h, int(option), C.byref(v)))
        return int(v.value)

    def msm_set_chunk(self, entries):
        """entries per accumulation lane for this context's MSMs (0 = automatic): short-lived workgroups next to a latency chain"""
        _chk(load().cg_msm_set_chunk(self.h, C.c_uint32(int(entries))))

    def close(self):
        if self.h:
            load().cg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _chk(load().cg_ctx_sync(self.h))

    @property
    def stream(self):
        return load().cg_ctx_stream(self.h)

    def set_stream(self, hip_stream):
        """launch on a stream owned by the host application (e.g. torch.cuda.Stream().cuda_stream)"""
        _chk(load().cg_ctx_set_stream(self.h, C.c_void_p(int(hip_stream))))

    # ---- memory
    def free_many(self, bufs):
        """several DevBufs released behind ONE release mark (cg_dev_free_many)"""
        live = [b for b in bufs if b.ptr]
        arr = (C.c_void_p * max(1, len(live)))(*[b.ptr for b in live])
        _chk(load().cg_dev_free_many(self.h, arr, C.c_size_t(len(live))))
        for b in live: b.ptr = 0

    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DevBuf(self, max(arr.nbytes, 16)).upload(arr)

    # ---- Rep3Rand's O(n) draws on the device (cg_chacha12_fr_rand_dev)
    def chacha12_fr_rand(self, curve, seed, word_pos, n):
        """n x F::rand over ChaCha12Rng::from_seed(seed) positioned at word_pos: (DevBuf of n x 32 B, word position afterwards)"""
        buf = self.alloc(max(n * 32, 16)); after = C.c_uint64(0)
        _chk(load().cg_chacha12_fr_rand_dev(self.h, curve, bytes(seed), C.c_uint64(word_pos), C.c_size_t(n), C.c_void_p(buf.ptr), C.byref(after)))
        return buf, after.value

    def chacha12_fr_rand_begin(self, curve, seed, word_pos, n):
        """the same draws enqueued and not waited for: (DevBuf, ticket for chacha12_fr_rand_finish)"""
        buf = self.alloc(max(n * 32, 16)); tk = C.c_int32(-1)
        _chk(load().cg_chacha12_fr_rand_dev_begin(self.h, curve, bytes(seed), C.c_uint64(word_pos), C.c_size_t(n), C.c_void_p(buf.ptr), C.byref(tk)))
        return buf, tk.value

    def chacha12_fr_rand_finish(self, ticket):
        """waits for the draw of that ticket: word position afterwards"""
        after = C.c_uint64(0)
        _chk(load().cg_chacha12_fr_rand_dev_finish(self.h, C.c_int32(ticket), C.byref(after)))
        return after.value

    # ---- page-locked staging + asynchronous copies (exchange chunks move under the compute)
    def host_alloc(self, shape, dtype=np.uint64):
        """page-locked numpy array (free with host_free)"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _chk(load().cg_host_alloc(C.c_size_t(n), C.byref(p)))
        arr = np.ctypeslib.as_array((C.c_uint8 * max(n, 1)).from_address(p.value))[:n].view(dtype).reshape(shape)
        return arr

    def host_free(self, arr): _chk(load().cg_host_free(C.c_void_p(arr.ctypes.data)))

    def download_begin(self, h_dst, d_src, nbytes=None, offset=0):
        t = C.c_int32(-1)
        _chk(load().cg_dev_download_begin(self.h, _hp(h_dst), C.c_void_p(_dp(d_src).value + offset), C.c_size_t(h_dst.nbytes if nbytes is None else nbytes), C.byref(t)))
        return t.value

    def stream_mark(self):
        """a point of the context's stream order (everything enqueued so far) that later downloads can be ordered behind"""
        m = C.c_int32(-1)
        _chk(load().cg_stream_mark(self.h, C.byref(m)))
        return int(m.value)

    def download_begin_after(self, h_dst, d_src, mark, nbytes=None, offset=0):
        """like download_begin, ordered behind `mark` instead of behind everything the stream holds now"""
        t = C.c_int32(-1)
        nb = h_dst.nbytes if nbytes is None else nbytes
        _chk(load().cg_dev_download_begin_after(self.h, C.c_void_p(h_dst.ctypes.data), C.c_void_p(_dp(d_src).value + offset), C.c_size_t(nb), int(mark), C.byref(t)))
        return int(t.value)

    def upload_begin(self, d_dst, h_src, after_stream=True, nbytes=None, offset=0):
        t = C.c_int32(-1)
        _chk(load().cg_dev_upload_begin(self.h, C.c_void_p(_dp(d_dst).value + offset), _hp(h_src), C.c_size_t(h_src.nbytes if nbytes is None else nbytes), int(bool(after_stream)), C.byref(t)))
        return t.value

    def copy_wait(self, ticket): _chk(load().cg_copy_wait(self.h, int(ticket)))
    def copy_fence(self, ticket): _chk(load().cg_copy_fence(self.h, int(ticket)))

    # ---- MSM
    def register_bases(self, curve, group, points, stride=None, infinity_offset=-1):
        pts = np.ascontiguousarray(points)
        rec = point_words(curve, group, 2) * 8
        stride = stride or rec
        n = pts.nbytes // stride
        h = C.c_void_p()
        _chk(load().cg_bases_register(self.h, curve, group, _hp(pts), C.c_size_t(n), C.c_size_t(stride), C.c_int64(infinity_offset), C.byref(h)))
        return Bases(self, curve, group, h, n)

    def register_bases_device(self, curve, group, d_points, n):
        h = C.c_void_p()
        _chk(load().cg_bases_register_device(self.h, curve, group, _dp(d_points), C.c_size_t(n), C.byref(h)))
        return Bases(self, curve, group, h, n)

    def check_on_curve(self, bases):
        """(number of non-infinity points off the curve, index of the first one or None) — device-side zkey validation"""
        nb, fb = C.c_uint64(0), C.c_uint64(0)
        _chk(load().cg_bases_check_on_curve(self.h, bases.h, C.byref(nb), C.byref(fb)))
        return nb.value, (None if fb.value == 2**64 - 1 else fb.value)

    def check_subgroup(self, bases):
        """(number of non-infinity points outside the prime-order subgroup, index of the first one or None); assumes on-curve points"""
        nb, fb = C.c_uint64(0), C.c_uint64(0)
        _chk(load().cg_bases_check_subgroup(self.h, bases.h, C.byref(nb), C.byref(fb)))
        return nb.value, (None if fb.value == 2**64 - 1 else fb.value)

    def precompute_bases(self, bases, c=20):
        """per-window precomputed tables (one-time, (254/c + 1) x memory): fewer mixed additions per point afterwards"""
        _chk(load().cg_bases_precompute(self.h, bases.h, int(c)))

    def synth_bases(self, curve, group, first, n):
        """device-side table [(first+i)*G] (bench / test tooling)"""
        h = C.c_void_p()
        _chk(load().cg_bases_synth_multiples(self.h, curve, group, C.c_uint64(first), C.c_size_t(n), C.byref(h)))
        return Bases(self, curve, group, h, n)

    def bases_from_scalars(self, curve, group, d_scalars, n):
        """device-side table [s_i * G] for n device-resident Montgomery scalars (setup tooling: synthetic CRS)"""
        h = C.c_void_p()
        _chk(load().cg_bases_from_scalars(self.h, curve, group, _dp(d_scalars), C.c_size_t(n), C.byref(h)))
        return Bases(self, curve, group, h, n)

    def bases_download(self, bases, offset, n):
        out = np.zeros((n, point_words(bases.curve, bases.group, 2)), dtype=np.uint64)
        _chk(load().cg_bases_download(self.h, bases.h, C.c_size_t(offset), C.c_size_t(n), _hp(out)))
        return out

    def msm(self, bases, scalars, offset=0, n=None):
        """host scalars: list of k arrays (n,4) uint64 -> (k, 3*coord_words) Jacobian"""
        sc = [np.ascontiguousarray(s, dtype=np.uint64) for s in scalars]
        n = sc[0].size // 4 if n is None else n
        k = len(sc)
        out = np.zeros((k, point_words(bases.curve, bases.group, 3)), dtype=np.uint64)
        ptrs = (C.c_void_p * k)(*[s.ctypes.data for s in sc])
        _chk(load().cg_msm(self.h, bases.h, C.c_size_t(offset), C.c_size_t(n), ptrs, k, _hp(out)))
        return out

    def msm_dev(self, bases, d_scalars, n, offset=0):
        k = len(d_scalars)
        out = np.zeros((k, point_words(bases.curve, bases.group, 3)), dtype=np.uint64)
        ptrs = (C.c_void_p * k)(*[_dp(s).value for s in d_scalars])
        _chk(load().cg_msm_dev(self.h, bases.h, C.c_size_t(offset), C.c_size_t(n), ptrs, k, _hp(out)))
        return out

    def msm_dev_begin(self, bases, d_scalars, n, offset=0):
        k = len(d_scalars)
        ptrs = (C.c_void_p * k)(*[_dp(s).value for s in d_scalars])
        t = C.c_int32(-1)
        _chk(load().cg_msm_dev_begin(self.h, bases.h, C.c_size_t(offset), C.c_size_t(n), ptrs, k, C.byref(t)))
        return (t.value, k, bases.curve, bases.group)

    def msm_dev_begin_multi(self, bases_list, d_scalars, n, offsets=None):
        """several tables x the same scalar vectors: one digit/sort schedule per vector shared by all tables"""
        nb, k = len(bases_list), len(d_scalars)
        tabs = (C.c_void_p * nb)(*[b.h.value for b in bases_list])
        offs = (C.c_size_t * nb)(*([0] * nb if offsets is None else offsets))
        ptrs = (C.c_void_p * k)(*[_dp(s).value for s in d_scalars])
        tk = (C.c_int32 * nb)()
        _chk(load().cg_msm_dev_begin_multi(self.h, nb, tabs, offs, C.c_size_t(n), ptrs, k, tk))
        return [(tk[i], k, bases_list[i].curve, bases_list[i].group) for i in range(nb)]

    def msm_end(self, ticket):
        t, k, curve, group = ticket
        out = np.zeros((k, point_words(curve, group, 3)), dtype=np.uint64)
        _chk(load().cg_msm_end(self.h, t, _hp(out)))
        return out

    def set_scatter_capacity(self, cap):
        """0 = automatic optimistic capacity, > 0 forced (tests), < 0 always the exact two-pass sort"""
        _chk(load().cg_msm_set_scatter_capacity(self.h, int(cap)))

    def set_msm_window(self, c):
        _chk(load().cg_msm_set_window(self.h, int(c)))

    # ---- NTT
    def ntt(self, curve, vecs, group_gen, inverse=False, coset_gen=None):
        """host vectors (list of (n,4) arrays), transformed copies are returned"""
        out = [np.ascontiguousarray(v, dtype=np.uint64).copy() for v in vecs]
        k = len(out)
        n = out[0].size // 4
        ptrs = (C.c_void_p * k)(*[v.ctypes.data for v in out])
        gg = np.ascontiguousarray(group_gen, dtype=np.uint64)
        cg = None if coset_gen is None else np.ascontiguousarray(coset_gen, dtype=np.uint64)
        _chk(load().cg_ntt(self.h, curve, ptrs, k, C.c_size_t(n), _hp(gg), int(inverse), _hp(cg)))
        return out

    def ntt_dev(self, curve, d_vecs, n, group_gen, inverse=False, coset_gen=None):
        k = len(d_vecs)
        ptrs = (C.c_void_p * k)(*[_dp(v).value for v in d_vecs])
        gg = np.ascontiguousarray(group_gen, dtype=np.uint64)
        cg = None if coset_gen is None else np.ascontiguousarray(coset_gen, dtype=np.uint64)
        _chk(load().cg_ntt_dev(self.h, curve, ptrs, k, C.c_size_t(n), _hp(gg), int(inverse), _hp(cg)))

    def ntt_coset_pair_dev(self, curve, d_vecs, n, group_gen, coset_gen):
        """v <- NTT(g^i * iNTT(v)_i) in one call (groth16.rs:175-188), natural order in and out"""
        k = len(d_vecs)
        ptrs = (C.c_void_p * k)(*[_dp(v).value for v in d_vecs])
        _chk(load().cg_ntt_coset_pair_dev(self.h, curve, ptrs, k, C.c_size_t(n), _hp(np.ascontiguousarray(group_gen, dtype=np.uint64)), _hp(np.ascontiguousarray(coset_gen, dtype=np.uint64))))

    # ---- vector ops (device operands)
    def vec_add(self, curve, out, a, b, n): _chk(load().cg_vec_add_dev(self.h, curve, _dp(out), _dp(a), _dp(b), C.c_size_t(n)))
    def vec_sub(self, curve, out, a, b, n): _chk(load().cg_vec_sub_dev(self.h, curve, _dp(out), _dp(a), _dp(b), C.c_size_t(n)))
    def vec_mul(self, curve, out, a, b, n): _chk(load().cg_vec_mul_dev(self.h, curve, _dp(out), _dp(a), _dp(b), C.c_size_t(n)))

    def vec_rep3_mul_local(self, curve, out, aa, ab, ba, bb, mask, n):
        _chk(load().cg_vec_rep3_mul_local_dev(self.h, curve, _dp(out), _dp(aa), _dp(ab), _dp(ba), _dp(bb), _dp(mask), C.c_size_t(n)))

    def vec_distribute_powers(self, curve, v, n, g, c):
        _chk(load().cg_vec_distribute_powers_dev(self.h, curve, _dp(v), C.c_size_t(n), _hp(np.ascontiguousarray(g, dtype=np.uint64)), _hp(np.ascontiguousarray(c, dtype=np.uint64))))

    def vec_affine(self, curve, out, a, n, c, d=None):
        _chk(load().cg_vec_affine_dev(self.h, curve, _dp(out), _dp(a), C.c_size_t(n), _hp(np.ascontiguousarray(c, dtype=np.uint64)),
                                      _hp(None if d is None else np.ascontiguousarray(d, dtype=np.uint64))))

    def vec_fill(self, curve, v, n, value): _chk(load().cg_vec_fill_dev(self.h, curve, _dp(v), C.c_size_t(n), _hp(np.ascontiguousarray(value, dtype=np.uint64))))
    def vec_gather_strided(self, curve, out, src, n, offset, stride): _chk(load().cg_vec_gather_strided_dev(self.h, curve, _dp(out), _dp(src), C.c_size_t(n), C.c_size_t(offset), C.c_size_t(stride)))
    def vec_lincomb(self, curve, out, out_off, out_stride, n, srcs, src_off, src_stride, coeffs):
        """out[out_off + i*out_stride] = sum_j coeffs[j] * srcs[j][src_off[j] + i*src_stride[j]] (element units, negative strides allowed)"""
        k = len(srcs)
        ptrs = (C.c_void_p * k)(*[_dp(x).value for x in srcs])
        offs = (C.c_int64 * k)(*[int(x) for x in src_off]); strides = (C.c_int64 * k)(*[int(x) for x in src_stride])
        _chk(load().cg_vec_lincomb_dev(self.h, curve, _dp(out), C.c_int64(out_off), C.c_int64(out_stride), C.c_size_t(n), k, ptrs, offs, strides,
                                       _hp(np.ascontiguousarray(coeffs, dtype=np.uint64))))

    def vec_prefix_prod(self, curve, out, src, n): _chk(load().cg_vec_prefix_prod_dev(self.h, curve, _dp(out), _dp(src), C.c_size_t(n)))
    def vec_prefix_sum(self, curve, out, src, n): _chk(load().cg_vec_prefix_sum_dev(self.h, curve, _dp(out), _dp(src), C.c_size_t(n)))
    def vec_inverse(self, curve, out, src, n): _chk(load().cg_vec_inverse_dev(self.h, curve, _dp(out), _dp(src), C.c_size_t(n)))

    def spmv_csr(self, curve, row_ptr, col, coeff, n_rows, pub, n_inputs, party, wit_a, wit_b, out_a, out_b):
        _chk(load().cg_spmv_csr_dev(self.h, curve, _dp(row_ptr), _dp(col), _dp(coeff), C.c_size_t(n_rows), _dp(pub), C.c_uint32(n_inputs), int(party),
                                    _dp(wit_a), _dp(wit_b), _dp(out_a), _dp(out_b)))

    # ---- host-buffer forms
    def vec_mul_host(self, curve, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.empty_like(a)
        _chk(load().cg_vec_mul(self.h, curve, _hp(out), _hp(a), _hp(b), C.c_size_t(a.size // 4)))
        return out

    def vec_rep3_mul_local_host(self, curve, aa, ab, ba, bb, mask=None):
        arrs = [np.ascontiguousarray(x, dtype=np.uint64) for x in (aa, ab, ba, bb)]
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint64)
        out = np.empty_like(arrs[0])
        _chk(load().cg_vec_rep3_mul_local(self.h, curve, _hp(out), *[_hp(x) for x in arrs], _hp(m), C.c_size_t(arrs[0].size // 4)))
        return out

    # ---- stats
    def stats_enable(self, on=True): _chk(load().cg_stats_enable(self.h, int(on)))

    def stats(self, reset=False):
        class ST(C.Structure):
            _fields_ = [("msm_ms", C.c_double), ("ntt_ms", C.c_double), ("vec_ms", C.c_double), ("spmv_ms", C.c_double),
                        ("msm_calls", C.c_uint64), ("ntt_calls", C.c_uint64), ("vec_calls", C.c_uint64), ("spmv_calls", C.c_uint64),
                        ("msm_sort_ms", C.c_double), ("msm_acc_g1_ms", C.c_double), ("msm_acc_g2_ms", C.c_double), ("msm_reduce_ms", C.c_double),
                        ("msm_sort_calls", C.c_uint64), ("msm_acc_g1_calls", C.c_uint64), ("msm_acc_g2_calls", C.c_uint64), ("msm_reduce_calls", C.c_uint64)]
        st = ST()
        _chk(load().cg_stats(self.h, C.byref(st), int(reset)))
        return {f: getattr(st, f) for f, _ in ST._fields_}


# ---- O(1) host helpers (no device needed) --------------------------------------------------------------
def dev_cache_trim(device=0):
    """give the device blocks parked by cg_dev_free back to the runtime (call when the device is idle); returns the bytes released"""
    n = C.c_size_t(0)
    _chk(load().cg_dev_cache_trim(int(device), C.byref(n)))
    return int(n.value)


def device_count():
    """HIP devices visible to the library (0 without a GPU)"""
    return int(load().cg_device_count())


def session_devices(world):
    """device list of a `world`-device session on THIS box: distinct GPUs as far as they exist, then repeats (a one-GPU box runs every
    context on GPU 0; on a node the hops between devices are real peer copies over xGMI)"""
    n = max(1, device_count())
    return [i % n for i in range(world)]


def set_option(option, value):
    """process-wide option of the hip library (cg_set_option)"""
    _chk(load().cg_set_option(int(option), C.c_int64(int(value))))


def get_option(option):
    v = C.c_int64(0)
    _chk(load().cg_get_option(int(option), C.byref(v)))
    return v.value


PREFLIGHT_ALLOW_SHARED, PREFLIGHT_ALLOW_STAGED = 1, 2


def device_preflight(devices, allow_shared=False, allow_staged=False):
    """cg_device_preflight: raises BackendError unless `devices` are distinct GPUs with peer access whose pairwise 1 MiB copies arrive intact;
    returns the report (bus ids, per-pair peer access and copy rate) as a dict"""
    import json
    devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    buf = C.create_string_buffer(1 << 16)
    _chk(load().cg_device_preflight(devs, len(devices), C.c_uint32((1 if allow_shared else 0) | (2 if allow_staged else 0)), buf, C.c_size_t(len(buf))))
    return json.loads(buf.value.decode())


def point_add(curve, group, a, b):
    out = np.zeros(point_words(curve, group, 3), dtype=np.uint64)
    _chk(load().cg_point_add(curve, group, _hp(np.ascontiguousarray(a)), _hp(np.ascontiguousarray(b)), _hp(out)))
    return out


def point_neg(curve, group, a):
    out = np.zeros(point_words(curve, group, 3), dtype=np.uint64)
    _chk(load().cg_point_neg(curve, group, _hp(np.ascontiguousarray(a)), _hp(out)))
    return out


def point_scalar_mul(curve, group, a, k):
    out = np.zeros(point_words(curve, group, 3), dtype=np.uint64)
    _chk(load().cg_point_scalar_mul(curve, group, _hp(np.ascontiguousarray(a)), _hp(np.ascontiguousarray(k)), _hp(out)))
    return out


class FixedBase:
    """8-bit window table of one base point (cg_fixed_base_*): host arithmetic, no device"""

    def __init__(self, curve, group, point_jacobian):
        self.curve, self.group, self.h = curve, group, C.c_void_p()
        p = np.ascontiguousarray(point_jacobian, dtype=np.uint64)
        _chk(load().cg_fixed_base_create(curve, group, _hp(p), C.byref(self.h)))

    def mul(self, k):
        out = np.zeros(point_words(self.curve, self.group, 3), dtype=np.uint64)
        _chk(load().cg_fixed_base_mul(self.h, _hp(np.ascontiguousarray(k, dtype=np.uint64)), _hp(out)))
        return out

    def close(self):
        if self.h: load().cg_fixed_base_destroy(self.h); self.h = None

    def __del__(self):
        try: self.close()
        except Exception: pass


def point_generator(curve, group):
    out = np.zeros(point_words(curve, group, 3), dtype=np.uint64)
    _chk(load().cg_point_generator(curve, group, _hp(out)))
    return out


def point_validate(curve, group, affine):
    """the reference's checks on a deserialised point (coordinates below the modulus, on the curve, in the subgroup); host arithmetic"""
    ok = C.c_int32(0)
    _chk(load().cg_point_validate(curve, group, _hp(np.ascontiguousarray(affine, dtype=np.uint64)), C.byref(ok)))
    return bool(ok.value)


def fr_is_canonical(curve, elements):
    e = np.ascontiguousarray(elements, dtype=np.uint64).reshape(-1, 4)
    ok = C.c_int32(0)
    _chk(load().cg_fr_is_canonical(curve, _hp(e), C.c_size_t(e.shape[0]), C.byref(ok)))
    return bool(ok.value)


def point_to_affine(curve, group, a):
    out = np.zeros(point_words(curve, group, 2), dtype=np.uint64)
    _chk(load().cg_point_to_affine(curve, group, _hp(np.ascontiguousarray(a)), _hp(out)))
    return out


def point_from_affine(curve, group, a):
    out = np.zeros(point_words(curve, group, 3), dtype=np.uint64)
    _chk(load().cg_point_from_affine(curve, group, _hp(np.ascontiguousarray(a)), _hp(out)))
    return out


def fr_op(curve, op, a, b=None):
    out = np.zeros(4, dtype=np.uint64)
    _chk(load().cg_fr_op(curve, {"add": 0, "sub": 1, "mul": 2, "inv": 3}[op], _hp(np.ascontiguousarray(a)), _hp(None if b is None else np.ascontiguousarray(b)), _hp(out)))
    return out


# ---- host-side mirror of the reference prover interface (libcogroth16_host.so, C++ over the C ABI) -----------------------
_host = None


def load_host():
    global _host
    if _host is None:
        load()
        if not os.path.exists(HOST_LIB_PATH):
            raise BackendError(f"{HOST_LIB_PATH} is missing: build it with `make -C collaborative-circom_amd/host`")
        _host = C.CDLL(HOST_LIB_PATH)
        _host.cgh_last_error.restype = C.c_char_p
    return _host


def _hchk(rc):
    if rc != 0:
        raise BackendError("cogroth16_host: " + load_host().cgh_last_error().decode())


def host_zkey_info(curve, path):
    info = (C.c_size_t * 7)()
    _hchk(load_host().cgh_zkey_info(curve, path.encode(), info))
    keys = ("n_vars", "n_public", "domain_size", "pow", "num_constraints", "nnz_a", "nnz_b")
    return dict(zip(keys, [int(x) for x in info]))


def host_zkey_validate(curve, path, device=0):
    """zkey -> device with the parser's per-point validation on the GPU; raises BackendError naming the first bad point.
    Returns (host seconds for read + decode, device seconds for upload + validation)."""
    secs = (C.c_double * 2)()
    _hchk(load_host().cgh_zkey_validate(int(device), curve, path.encode(), secs))
    return float(secs[0]), float(secs[1])


def host_proof_to_json(curve, proof):
    """Groth16Proof JSON text (proof.rs:8-29) of a packed proof A || B || C"""
    buf = C.create_string_buffer(4096)
    _hchk(load_host().cgh_proof_to_json(curve, _hp(np.ascontiguousarray(proof, dtype=np.uint64)), buf, C.c_size_t(4096)))
    return buf.value.decode()


def host_proof_from_json(curve, text):
    nq = 6 if curve == BLS12_381 else 4
    out = np.zeros(8 * nq, dtype=np.uint64)
    _hchk(load_host().cgh_proof_from_json(curve, text.encode(), _hp(out)))
    return out


def host_public_to_json(curve, pub):
    """public.json text (co-circom.rs:620-628) of n public signals in Montgomery form (without the leading constant 1)"""
    pub = np.ascontiguousarray(pub, dtype=np.uint64).reshape(-1, 4)
    buf = C.create_string_buffer(96 * max(1, pub.shape[0]) + 16)
    _hchk(load_host().cgh_public_to_json(curve, _hp(pub), C.c_size_t(pub.shape[0]), buf, C.c_size_t(len(buf))))
    return buf.value.decode()


def prove_shamir(curve, zkey_path, n, t, pub, wits, streams, device=0, want_h=False, preprocess=0):
    """n Shamir parties (threads) with threshold t on one GPU; returns (n proofs[, party 0's h shares]).  preprocess = number of
    secrets every party double-shares up front on the GPU (ShamirProtocol::preprocess); 0 = the reference's lazy batches of 1024."""
    info = host_zkey_info(curve, zkey_path)
    nq = 6 if curve == BLS12_381 else 4
    out = np.zeros((n, 8 * nq), dtype=np.uint64)
    h = np.zeros((info["domain_size"], 4), dtype=np.uint64) if want_h else None
    keep = [[np.ascontiguousarray(x, dtype=np.uint64) for x in lst] for lst in (wits, streams)]
    arr = lambda lst: (C.c_void_p * n)(*[x.ctypes.data for x in lst])
    _hchk(load_host().cgh_prove_shamir(int(device), curve, zkey_path.encode(), int(n), int(t), _hp(np.ascontiguousarray(pub, dtype=np.uint64)),
                                       arr(keep[0]), arr(keep[1]), C.c_size_t(keep[1][0].shape[0]), C.c_size_t(int(preprocess)), _hp(out), _hp(h) if want_h else None))
    return (out, h) if want_h else out


def host_synth_circuit(curve, log_m, seed, zkey_path, wtns_path, device=0):
    """synthetic satisfiable circuit of 2^log_m - 2 constraints with a valid Groth16 CRS (toxic waste from `seed`, point tables by
    fixed-base batch multiplication on the GPU), written as snarkjs-format .zkey + .wtns files (bench / test tooling)"""
    _hchk(load_host().cgh_synth_circuit(int(device), curve, int(log_m), C.c_uint64(seed), zkey_path.encode(), wtns_path.encode()))


def host_set_zkey_validation(on):
    """the prove entry points and session_open validate every zkey point on the GPU by default (the reference's parser does);
    callers that validated the file before can switch it off"""
    _hchk(load_host().cgh_set_zkey_validation(int(bool(on))))


# process-wide options of the host library (include/cogroth16_host.h: cgh_set_option)
HOST_OPT_XCHG_ASYNC_MIN, HOST_OPT_DEVICE_MASKS_MIN, HOST_OPT_XCHG_COPY_STREAM_MIN, HOST_OPT_SECOND_CONTEXT_MIN_LOG, HOST_OPT_DISTRIBUTED_MAP, HOST_OPT_ONE_CONTEXT, HOST_OPT_SPLIT_FIRST_MSM_MIN, HOST_OPT_CTX_WIDE_LOG, HOST_OPT_CTX_OFF_MAIN_LOG, HOST_OPT_CTX_SOLO_LOG = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10


def host_set_option(option, value):
    _hchk(load_host().cgh_set_option(int(option), C.c_int64(int(value))))


def host_get_option(option):
    v = C.c_int64(0)
    _hchk(load_host().cgh_get_option(int(option), C.byref(v)))
    return v.value


class host_options:
    """with cg.host_options({cg.HOST_OPT_XCHG_ASYNC_MIN: 4096}): ... — set for the block, restored afterwards (tests)"""

    def __init__(self, values):
        self.values, self.saved = dict(values), {}

    def __enter__(self):
        for k, v in self.values.items():
            self.saved[k] = host_get_option(k); host_set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            host_set_option(k, v)
        return False


class ProvingSession:
    """zkey read, uploaded and (optionally) given per-window precomputed tables once; proofs then cost what co-circom.rs:503-506 times"""

    def __init__(self, curve, zkey_path, precompute=True, device=0, validate=True, devices=None, additive_h=False, shared_devices=False):
        """devices: several GPUs of one node for this party (cgh_session_open_multi): devices[0] runs the witness map and slice 0 of
        every MSM, devices[i] slice i.  additive_h: REP3 proofs run the additive-quotient variant (CGH_SESSION_ADDITIVE_H, opt-in: not the
        reference's message sequence, same proof).  shared_devices: the list may name one GPU several times (CGH_SESSION_SHARED_DEVICES:
        one-GPU tests and planning runs) — without it a multi-device session opens only on DISTINCT GPUs that pass cg_device_preflight"""
        self.curve, self.info = curve, host_zkey_info(curve, zkey_path)
        h = C.c_void_p()
        devs = [int(device)] if devices is None else [int(d) for d in devices]
        arr = (C.c_int32 * len(devs))(*devs)
        _hchk(load_host().cgh_session_open_multi(arr, len(devs), curve, zkey_path.encode(), -1 if precompute is True else int(precompute), C.c_uint32((0 if validate else 1) | (2 if additive_h else 0) | (4 if shared_devices else 0)), C.byref(h)))
        self.h = h

    def close(self):
        if self.h: load_host().cgh_session_close(self.h); self.h = None

    def prove_plain(self, witness, r, s):
        """returns (proof, seconds)"""
        nq = 6 if self.curve == BLS12_381 else 4
        out = np.zeros(8 * nq, dtype=np.uint64); sec = (C.c_double * 1)()
        _hchk(load_host().cgh_session_prove_plain(self.h, _hp(np.ascontiguousarray(witness, dtype=np.uint64)), _hp(np.ascontiguousarray(r, dtype=np.uint64)),
                                                  _hp(np.ascontiguousarray(s, dtype=np.uint64)), _hp(out), sec))
        return out, sec[0]

    def prove_rep3(self, pub, wit_a, wit_b, streams, solo=True):
        """three co-located parties; returns (3 proofs, seconds of the three together, seconds of party 0 replayed alone on the GPU)"""
        nq = 6 if self.curve == BLS12_381 else 4
        out = np.zeros((3, 8 * nq), dtype=np.uint64); sec = (C.c_double * 2)()
        keep = [[np.ascontiguousarray(x, dtype=np.uint64) for x in lst] for lst in (wit_a, wit_b, streams)]
        arr = lambda lst: (C.c_void_p * 3)(*[x.ctypes.data for x in lst])
        _hchk(load_host().cgh_session_prove_rep3(self.h, _hp(np.ascontiguousarray(pub, dtype=np.uint64)), arr(keep[0]), arr(keep[1]), arr(keep[2]),
                                                 C.c_size_t(keep[2][0].shape[0]), _hp(out), sec if solo else None))
        return out, sec[0], sec[1]


# ---- one REP3 party with the caller's network and randomness (cgh_session_prove_rep3_party) ------------------------------------
_SEND = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t)
_RECV = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t)
_RECV_PINNED = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class Rep3NetTable(C.Structure):
    """cgh_rep3_net (include/cogroth16_host.h): the party's channels to its two peers"""
    _fields_ = [("user", C.c_void_p), ("party_id", C.c_int32), ("send_next", _SEND), ("recv_prev", _RECV), ("send_prev", _SEND), ("recv_next", _RECV),
                ("recv_prev_pinned", _RECV_PINNED)]


_MASKS = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p))
_FES = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)
_EC = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_void_p)


class Rep3RandTable(C.Structure):
    """cgh_rep3_rand: the party's correlated randomness (Rep3Rand, rep3/rngs.rs:25-62)"""
    _fields_ = [("user", C.c_void_p), ("masking_field_elements", _MASKS), ("random_fes", _FES), ("masking_ec_element", _EC)]


class LoopbackHub:
    """three parties of one process joined by in-memory queues (cgh_loopback_*); net(i) is party i's callback table"""

    def __init__(self):
        self.h = C.c_void_p(); _hchk(load_host().cgh_loopback_create(C.byref(self.h)))

    def net(self, party, record=False):
        t = Rep3NetTable(); _hchk(load_host().cgh_loopback_net(self.h, int(party), int(bool(record)), C.byref(t))); return t

    def replay_net(self, party):
        t = Rep3NetTable(); _hchk(load_host().cgh_loopback_replay_net(self.h, int(party), C.byref(t))); return t

    def abort(self): load_host().cgh_loopback_abort(self.h)

    def close(self):
        if self.h: load_host().cgh_loopback_destroy(self.h); self.h = None


class StreamRand:
    """Rep3Rand over two pre-generated streams of field elements (cgh_stream_rand_create); .table is the callback table"""

    def __init__(self, curve, rng1, rng2):
        self.keep = (np.ascontiguousarray(rng1, dtype=np.uint64), np.ascontiguousarray(rng2, dtype=np.uint64))
        self.h = C.c_void_p(); self.table = Rep3RandTable()
        _hchk(load_host().cgh_stream_rand_create(curve, _hp(self.keep[0]), _hp(self.keep[1]), C.c_size_t(self.keep[0].shape[0]), C.byref(self.h), C.byref(self.table)))

    def close(self):
        if self.h: load_host().cgh_stream_rand_destroy(self.h); self.h = None


_CH_STATE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))
_CH_SETPOS = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64)


class Rep3ChaChaTable(C.Structure):
    """cgh_rep3_chacha: seeds and word positions of Rep3Rand's two ChaCha12 generators, so that the masking vectors are drawn on the GPU"""
    _fields_ = [("user", C.c_void_p), ("get_state", _CH_STATE), ("set_word_pos", _CH_SETPOS)]


class ChaChaRand:
    """Rep3Rand over two ChaCha12 generators (cgh_chacha_rand_create): .table = the O(1) / host callbacks, .streams = the generator description"""

    def __init__(self, curve, seed1, seed2):
        self.h = C.c_void_p(); self.table = Rep3RandTable(); self.streams = Rep3ChaChaTable()
        _hchk(load_host().cgh_chacha_rand_create(curve, bytes(seed1), bytes(seed2), C.byref(self.h), C.byref(self.table), C.byref(self.streams)))

    def positions(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        _hchk(load_host().cgh_chacha_rand_positions(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self.h: load_host().cgh_chacha_rand_destroy(self.h); self.h = None


def chacha12_fr_rand_host(curve, seed, word_pos, n):
    """n x F::rand over ChaCha12Rng::from_seed(seed) at word_pos, on the host (one thread): (n x 4 limbs, word position afterwards)"""
    out = np.zeros((n, 4), dtype=np.uint64); after = C.c_uint64(0)
    _hchk(load_host().cgh_chacha12_fr_rand_host(curve, bytes(seed), C.c_uint64(word_pos), C.c_size_t(n), _hp(out), C.byref(after)))
    return out, after.value


_SH_SEND = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t)
_SH_RECV = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t)
_SH_RAND = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p)


class ShamirNetTable(C.Structure):
    """cgh_shamir_net: any-to-any channels of one of n parties (shamir/network.rs:17-59)"""
    _fields_ = [("user", C.c_void_p), ("party_id", C.c_int32), ("num_parties", C.c_int32), ("send", _SH_SEND), ("recv", _SH_RECV)]


class ShamirRandTable(C.Structure):
    """cgh_shamir_rand: the party's private RNG (F::rand draws in the reference's order)"""
    _fields_ = [("user", C.c_void_p), ("random_field_elements", _SH_RAND)]


class ShamirLoopbackHub:
    """n Shamir parties of one process joined by in-memory queues (cgh_shamir_loopback_*); net(i) is party i's callback table"""

    def __init__(self, n):
        self.h = C.c_void_p(); _hchk(load_host().cgh_shamir_loopback_create(int(n), C.byref(self.h)))

    def net(self, party, record=False):
        t = ShamirNetTable(); _hchk(load_host().cgh_shamir_loopback_net(self.h, int(party), int(bool(record)), C.byref(t))); return t

    def replay_net(self, party):
        t = ShamirNetTable(); _hchk(load_host().cgh_shamir_loopback_replay_net(self.h, int(party), C.byref(t))); return t

    def abort(self): load_host().cgh_shamir_loopback_abort(self.h)

    def close(self):
        if self.h: load_host().cgh_shamir_loopback_destroy(self.h); self.h = None


def host_prove_shamir_party(session, threshold, pub, wit, net_table, rand_table, preprocess=0):
    """ONE Shamir party on an open ProvingSession through the callback ABI; returns (proof, seconds).  Call it from one thread per party."""
    nq = 6 if session.curve == BLS12_381 else 4
    out = np.zeros(8 * nq, dtype=np.uint64); sec = (C.c_double * 1)()
    keep = [np.ascontiguousarray(x, dtype=np.uint64) for x in (pub, wit)]
    _hchk(load_host().cgh_session_prove_shamir_party(session.h, int(threshold), _hp(keep[0]), _hp(keep[1]), C.byref(net_table), C.byref(rand_table),
                                                     C.c_size_t(int(preprocess)), _hp(out), sec))
    return out, sec[0]


def host_prove_shamir_party_seeded(session, threshold, pub, wit, net_table, seed, preprocess=0):
    """ONE Shamir party whose private randomness is a ChaCha12 generator run by the library from `seed` (32 bytes); returns (proof, seconds)"""
    nq = 6 if session.curve == BLS12_381 else 4
    out = np.zeros(8 * nq, dtype=np.uint64); sec = (C.c_double * 1)()
    keep = [np.ascontiguousarray(x, dtype=np.uint64) for x in (pub, wit)]
    _hchk(load_host().cgh_session_prove_shamir_party_seeded(session.h, int(threshold), _hp(keep[0]), _hp(keep[1]), C.byref(net_table), bytes(seed),
                                                            C.c_size_t(int(preprocess)), _hp(out), sec))
    return out, sec[0]


def host_prove_rep3_party(session, pub, wit_a, wit_b, net_table, rand_table, streams_table=None):
    """ONE REP3 party on an open ProvingSession through the callback ABI; returns (proof, seconds).  Call it from one thread per party.
    streams_table (Rep3ChaChaTable): the masking vectors are drawn on the GPU from the described ChaCha12 generators."""
    nq = 6 if session.curve == BLS12_381 else 4
    out = np.zeros(8 * nq, dtype=np.uint64); sec = (C.c_double * 1)()
    keep = [np.ascontiguousarray(x, dtype=np.uint64) for x in (pub, wit_a, wit_b)]
    _hchk(load_host().cgh_session_prove_rep3_party_ex(session.h, _hp(keep[0]), _hp(keep[1]), _hp(keep[2]), C.byref(net_table), C.byref(rand_table),
                                                      C.byref(streams_table) if streams_table is not None else None, _hp(out), sec))
    return out, sec[0]


def host_plonk_zkey_info(curve, path):
    info = (C.c_size_t * 6)()
    _hchk(load_host().cgh_plonk_zkey_info(curve, path.encode(), info))
    return dict(zip(("n_vars", "n_public", "domain_size", "power", "n_additions", "n_constraints"), [int(x) for x in info]))


PLONK_COMMITS = ("a", "b", "c", "z", "t1", "t2", "t3", "wxi", "wxiw")
PLONK_CHALLENGES = ("beta", "gamma", "alpha", "xi", "v")
PLONK_EVALS = ("eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw")


def _pad_blind(blind):
    blind = np.ascontiguousarray(blind, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((11, 4), dtype=np.uint64); out[:blind.shape[0]] = blind
    return out


def plonk_prove_plain(curve, zkey_path, full_witness, blind, upto=5, device=0, want_t=False, want_poly_z=False):
    """co-plonk with the plain driver on the GPU through round `upto` (<= 5): dict of commitments, challenges, evaluations[, t polys, poly_z]
    (the proof of co-plonk/src/plonk.rs = the nine commitments and six evaluations)"""
    info = host_plonk_zkey_info(curve, zkey_path)
    nq = 6 if curve == BLS12_381 else 4; n = info["domain_size"]
    commits = np.zeros((9, 2 * nq), dtype=np.uint64); ch = np.zeros((5, 4), dtype=np.uint64); ev = np.zeros((6, 4), dtype=np.uint64)
    tp = np.zeros((3 * n + 8, 4), dtype=np.uint64) if want_t else None
    pz = np.zeros((n + 3, 4), dtype=np.uint64) if want_poly_z else None
    _hchk(load_host().cgh_plonk_prove_plain(int(device), curve, zkey_path.encode(), _hp(np.ascontiguousarray(full_witness, dtype=np.uint64)),
                                            _hp(_pad_blind(blind)), int(upto), _hp(commits), _hp(ch), _hp(ev), _hp(tp) if want_t else None, _hp(pz) if want_poly_z else None))
    out = dict(zip(PLONK_COMMITS, commits)); out.update(zip(PLONK_CHALLENGES, ch)); out.update(zip(PLONK_EVALS, ev))
    if want_t: out.update(t1_poly=tp[:n + 1], t2_poly=tp[n + 1:2 * n + 2], t3_poly=tp[2 * n + 2:])
    if want_poly_z: out["poly_z"] = pz
    return out


def plonk_round1_plain(curve, zkey_path, full_witness, blind, device=0):
    """the three wire commitments (3, packed G1)"""
    r = plonk_prove_plain(curve, zkey_path, full_witness, blind, upto=1, device=device)
    return np.stack([r["a"], r["b"], r["c"]])


def plonk_round2_plain(curve, zkey_path, full_witness, blind, device=0, want_poly=False):
    """rounds 1 + 2: (beta, gamma, commit_z[, poly_z])"""
    r = plonk_prove_plain(curve, zkey_path, full_witness, blind, upto=2, device=device, want_poly_z=want_poly)
    return (r["beta"], r["gamma"], r["z"], r["poly_z"]) if want_poly else (r["beta"], r["gamma"], r["z"])


def plonk_prove_rep3(curve, zkey_path, pub, wit_a, wit_b, blind_a, blind_b, streams=None, upto=5, device=0):
    """three REP3 parties (threads) on one GPU through round `upto`; returns a list of three dicts (one per party) like plonk_prove_plain"""
    nq = 6 if curve == BLS12_381 else 4
    commits = np.zeros((3, 9, 2 * nq), dtype=np.uint64); ch = np.zeros((3, 5, 4), dtype=np.uint64); ev = np.zeros((3, 6, 4), dtype=np.uint64)
    lists = [wit_a, wit_b, [_pad_blind(x) for x in blind_a], [_pad_blind(x) for x in blind_b]] + ([streams] if streams is not None else [])
    keep = [[np.ascontiguousarray(x, dtype=np.uint64) for x in lst] for lst in lists]
    arr = lambda lst: (C.c_void_p * 3)(*[x.ctypes.data for x in lst])
    _hchk(load_host().cgh_plonk_prove_rep3(int(device), curve, zkey_path.encode(), _hp(np.ascontiguousarray(pub, dtype=np.uint64)), arr(keep[0]), arr(keep[1]),
                                           arr(keep[2]), arr(keep[3]), arr(keep[4]) if streams is not None else None,
                                           C.c_size_t(keep[4][0].shape[0] if streams is not None else 0), int(upto), _hp(commits), _hp(ev), _hp(ch)))
    out = []
    for i in range(3):
        dct = dict(zip(PLONK_COMMITS, commits[i])); dct.update(zip(PLONK_CHALLENGES, ch[i])); dct.update(zip(PLONK_EVALS, ev[i]))
        out.append(dct)
    return out


def plonk_prove_rep3_party(curve, zkey_path, pub, wit_a, wit_b, net_table, rand_table, blind_a=None, blind_b=None, upto=5, device=0, streams_table=None):
    """ONE REP3 party of co-plonk through the callback ABI (cgh_plonk_prove_rep3_party); blind_a/blind_b None = drawn with rand() first.
    Returns a dict like plonk_prove_plain.  Call it from one thread per party."""
    nq = 6 if curve == BLS12_381 else 4
    commits = np.zeros((9, 2 * nq), dtype=np.uint64); ch = np.zeros((5, 4), dtype=np.uint64); ev = np.zeros((6, 4), dtype=np.uint64)
    keep = [np.ascontiguousarray(x, dtype=np.uint64) for x in (pub, wit_a, wit_b)]
    bl = [None if x is None else _pad_blind(x) for x in (blind_a, blind_b)]
    _hchk(load_host().cgh_plonk_prove_rep3_party_ex(int(device), curve, zkey_path.encode(), _hp(keep[0]), _hp(keep[1]), _hp(keep[2]),
                                                    None if bl[0] is None else _hp(bl[0]), None if bl[1] is None else _hp(bl[1]),
                                                    C.byref(net_table), C.byref(rand_table), C.byref(streams_table) if streams_table is not None else None,
                                                    int(upto), _hp(commits), _hp(ev), _hp(ch)))
    dct = dict(zip(PLONK_COMMITS, commits)); dct.update(zip(PLONK_CHALLENGES, ch)); dct.update(zip(PLONK_EVALS, ev))
    return dct


def plonk_prove_shamir(curve, zkey_path, n, t, pub, wits, blinds, streams, upto=5, device=0):
    """n Shamir parties (threshold t) on one GPU through round `upto`; returns a list of n dicts like plonk_prove_plain"""
    nq = 6 if curve == BLS12_381 else 4
    commits = np.zeros((n, 9, 2 * nq), dtype=np.uint64); ch = np.zeros((n, 5, 4), dtype=np.uint64); ev = np.zeros((n, 6, 4), dtype=np.uint64)
    keep = [[np.ascontiguousarray(x, dtype=np.uint64) for x in lst] for lst in (wits, [_pad_blind(x) for x in blinds], streams)]
    arr = lambda lst: (C.c_void_p * n)(*[x.ctypes.data for x in lst])
    _hchk(load_host().cgh_plonk_prove_shamir(int(device), curve, zkey_path.encode(), int(n), int(t), _hp(np.ascontiguousarray(pub, dtype=np.uint64)),
                                             arr(keep[0]), arr(keep[1]), arr(keep[2]), C.c_size_t(keep[2][0].shape[0]), int(upto), _hp(commits), _hp(ev), _hp(ch)))
    out = []
    for i in range(n):
        dct = dict(zip(PLONK_COMMITS, commits[i])); dct.update(zip(PLONK_CHALLENGES, ch[i])); dct.update(zip(PLONK_EVALS, ev[i]))
        out.append(dct)
    return out


def plonk_round1_rep3(curve, zkey_path, pub, wit_a, wit_b, blind_a, blind_b, device=0):
    """round 1 only: (3 parties, 3 commitments, packed G1)"""
    r = plonk_prove_rep3(curve, zkey_path, pub, wit_a, wit_b, blind_a, blind_b, streams=None, upto=1, device=device)
    return np.stack([np.stack([p["a"], p["b"], p["c"]]) for p in r])


def host_shared_witness_write(curve, path, pub, a, b=None):
    """`.shared` witness file (bincode + ark-compressed vectors): REP3 when b is given, Shamir otherwise.  Layout restated from the
    reference's types; the snapshot holds no .shared fixture to pin it against."""
    pub = np.ascontiguousarray(pub, dtype=np.uint64).reshape(-1, 4); a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    _hchk(load_host().cgh_shared_witness_write(curve, path.encode(), 0 if b is not None else 1, _hp(pub), C.c_size_t(pub.shape[0]), _hp(a), _hp(bb), C.c_size_t(a.shape[0])))


def host_shared_witness_read(curve, path, rep3=True):
    sizes = (C.c_size_t * 2)()
    _hchk(load_host().cgh_shared_witness_read(curve, path.encode(), 0 if rep3 else 1, sizes, None, None, None))
    pub = np.zeros((sizes[0], 4), dtype=np.uint64); a = np.zeros((sizes[1], 4), dtype=np.uint64); b = np.zeros((sizes[1], 4), dtype=np.uint64) if rep3 else None
    _hchk(load_host().cgh_shared_witness_read(curve, path.encode(), 0 if rep3 else 1, sizes, _hp(pub), _hp(a), _hp(b)))
    return (pub, a, b) if rep3 else (pub, a)


def host_plonk_proof_to_json(curve, proof):
    """PlonkProof JSON text (circom-types/src/plonk/proof.rs) of a proof dict (PLONK_COMMITS + PLONK_EVALS keys)"""
    commits = np.ascontiguousarray(np.stack([proof[k] for k in PLONK_COMMITS]), dtype=np.uint64)
    evals = np.ascontiguousarray(np.stack([proof[k] for k in PLONK_EVALS]), dtype=np.uint64)
    buf = C.create_string_buffer(8192)
    _hchk(load_host().cgh_plonk_proof_to_json(curve, _hp(commits), _hp(evals), buf, C.c_size_t(8192)))
    return buf.value.decode()


def host_plonk_proof_from_json(curve, text):
    nq = 6 if curve == BLS12_381 else 4
    commits = np.zeros((9, 2 * nq), dtype=np.uint64); evals = np.zeros((6, 4), dtype=np.uint64)
    _hchk(load_host().cgh_plonk_proof_from_json(curve, text.encode(), _hp(commits), _hp(evals)))
    out = dict(zip(PLONK_COMMITS, commits)); out.update(zip(PLONK_EVALS, evals))
    return out


def host_plonk_transcript(curve, items):
    """items: list of ("scalar", limbs) / ("point", packed G1 limbs) -> challenge computed by the host mirror's Keccak256 transcript"""
    n = len(items)
    kinds = (C.c_int32 * n)(*[0 if k == "scalar" else 1 for k, _ in items])
    keep = [np.ascontiguousarray(v, dtype=np.uint64) for _, v in items]
    ptrs = (C.c_void_p * n)(*[v.ctypes.data for v in keep])
    out = np.zeros(4, dtype=np.uint64)
    _hchk(load_host().cgh_plonk_transcript(curve, kinds, ptrs, n, _hp(out)))
    return out


def host_read_wtns(curve, path):
    n = C.c_size_t(0)
    _hchk(load_host().cgh_read_wtns(curve, path.encode(), None, C.c_size_t(0), C.byref(n)))
    out = np.zeros((n.value, 4), dtype=np.uint64)
    _hchk(load_host().cgh_read_wtns(curve, path.encode(), _hp(out), C.c_size_t(n.value), C.byref(n)))
    return out


def prove_plain(curve, zkey_path, full_witness, r, s, device=0, want_h=False):
    """PlainHipDriver + CoGroth16::prove; returns packed proof A||B||C (and the h vector)"""
    info = host_zkey_info(curve, zkey_path)
    w = np.ascontiguousarray(full_witness, dtype=np.uint64)
    out = np.zeros(8 * fq_limbs(curve), dtype=np.uint64)
    h = np.zeros((info["domain_size"], 4), dtype=np.uint64) if want_h else None
    _hchk(load_host().cgh_prove_plain(int(device), curve, zkey_path.encode(), _hp(w), _hp(np.ascontiguousarray(r, dtype=np.uint64)),
                                      _hp(np.ascontiguousarray(s, dtype=np.uint64)), _hp(out), _hp(h)))
    return (out, h) if want_h else out


def prove_rep3(curve, zkey_path, pub, wit_a, wit_b, streams, device=0, want_h=False):
    """three Rep3HipProtocol parties on three threads over the in-process network; returns the three proofs"""
    info = host_zkey_info(curve, zkey_path)
    pub = np.ascontiguousarray(pub, dtype=np.uint64)
    wa = [np.ascontiguousarray(x, dtype=np.uint64) for x in wit_a]
    wb = [np.ascontiguousarray(x, dtype=np.uint64) for x in wit_b]
    st = [np.ascontiguousarray(x, dtype=np.uint64) for x in streams]
    arr = lambda xs: (C.c_void_p * 3)(*[x.ctypes.data for x in xs])
    out = np.zeros((3, 8 * fq_limbs(curve)), dtype=np.uint64)
    h = np.zeros((2, info["domain_size"], 4), dtype=np.uint64) if want_h else None
    _hchk(load_host().cgh_prove_rep3(int(device), curve, zkey_path.encode(), _hp(pub), arr(wa), arr(wb), arr(st), C.c_size_t(st[0].shape[0]), _hp(out), _hp(h)))
    return (out, h) if want_h else out
