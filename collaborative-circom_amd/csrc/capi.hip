// C ABI of the gfx950 co-groth16 backend (declared in include/cogroth16_hip.h).  Host-side launch logic only:
// every O(n) computation happens in the kernels of vec_kernels.hpp / ntt_kernels.hpp / msm_kernels.hpp.
// There is deliberately no CPU fallback here: without a HIP device cg_ctx_create fails.
#include "capi_internal.hpp"


template <int OP>
int32_t vec_binary(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_a, const void* d_b, size_t n) {
    if (!ctx || !d_out || !d_a || !d_b) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        StatScope ss(ctx, TAG_VEC);
        return launch_vec_binary<Fr>(ctx->stream, OP, (Fr*)d_out, (const Fr*)d_a, (const Fr*)d_b, n);
    });
}

// curve coefficient b of y^2 = x^3 + b for the group's coordinate field, Montgomery form
template <class F> struct CurveB;
template <> struct CurveB<Bn254Fq> { static Bn254Fq get() { Bn254Fq t = Bn254Fq::one(); return t + t + t; } };
template <> struct CurveB<Fp2<Bn254Fq>> { static Fp2<Bn254Fq> get() {   // 3 / (9 + u)
    Bn254Fq one = Bn254Fq::one(), three = one + one + one, nine = three + three + three;
    Fp2<Bn254Fq> xi = {nine, one}; Fp2<Bn254Fq> inv = fp_inverse(xi); return {inv.c0 * three, inv.c1 * three}; } };
#if CG_WITH_BLS
template <> struct CurveB<Bls381Fq> { static Bls381Fq get() { Bls381Fq t = Bls381Fq::one(); t = t + t; return t + t; } };
template <> struct CurveB<Fp2<Bls381Fq>> { static Fp2<Bls381Fq> get() { Bls381Fq f = CurveB<Bls381Fq>::get(); return {f, f}; } };
#endif

// ---- constants of the endomorphism-based subgroup tests (subgroup.hpp), computed once per process on the host
namespace {
// (p - 1) / d for the coordinate field's modulus, little-endian 32-bit limbs (d = 2 or 3 divides p - 1 for both curves)
template <class B> void modulus_minus_one_over(uint32_t d, uint32_t (&out)[B::N]) {
    uint32_t t[B::N]; for (int i = 0; i < B::N; i++) t[i] = B::Params::P[i];
    t[0] -= 1;                                                                       // p is odd: no borrow
    uint64_t rem = 0;
    for (int i = B::N - 1; i >= 0; i--) { const uint64_t cur = (rem << 32) | t[i]; out[i] = (uint32_t)(cur / d); rem = cur % d; }
}
template <class F> Affine<F> group_generator();
template <> Affine<Fp2<Bn254Fq>> group_generator() { Affine<Fp2<Bn254Fq>> a; memcpy(&a, Bn254G2_GEN, sizeof a); return a; }
#if CG_WITH_BLS
template <> Affine<Fp2<Bls381Fq>> group_generator() { Affine<Fp2<Bls381Fq>> a; memcpy(&a, Bls381G2_GEN, sizeof a); return a; }
template <> Affine<Bls381Fq> group_generator() { Affine<Bls381Fq> a; memcpy(&a, Bls381G1_GEN, sizeof a); return a; }
#endif
// psi constants: xi^((p-1)/3), xi^((p-1)/2) for a D-type twist, their inverses for an M-type twist; the generator decides
template <class B> FastSubgroup<Fp2<B>> make_psi_subgroup(const Fp2<B>& xi) {
    uint32_t e3[B::N], e2[B::N];
    modulus_minus_one_over<B>(3, e3); modulus_minus_one_over<B>(2, e2);
    const Fp2<B> gx = fp_pow(xi, e3, B::N), gy = fp_pow(xi, e2, B::N);
    const Affine<Fp2<B>> gen = group_generator<Fp2<B>>();
    FastSubgroup<Fp2<B>> c;
    c.psi = PsiMap<B>{gx, gy};
    if (c.contains(gen)) return c;
    c.psi = PsiMap<B>{fp_inverse(gx), fp_inverse(gy)};
    if (c.contains(gen)) return c;
    throw std::runtime_error("subgroup test constants: the generator fails both twist conventions");
}
template <class F> struct FastSubgroupFactory { static FastSubgroup<F> make() { return FastSubgroup<F>(); } };
template <> struct FastSubgroupFactory<Fp2<Bn254Fq>> { static FastSubgroup<Fp2<Bn254Fq>> make() {
    Bn254Fq one = Bn254Fq::one(), three = one + one + one, nine = three + three + three;
    return make_psi_subgroup<Bn254Fq>(Fp2<Bn254Fq>{nine, one}); } };                 // xi = 9 + u
#if CG_WITH_BLS
template <> struct FastSubgroupFactory<Fp2<Bls381Fq>> { static FastSubgroup<Fp2<Bls381Fq>> make() {
    return make_psi_subgroup<Bls381Fq>(Fp2<Bls381Fq>{Bls381Fq::one(), Bls381Fq::one()}); } };   // xi = 1 + u
template <> struct FastSubgroupFactory<Bls381Fq> { static FastSubgroup<Bls381Fq> make() {
    uint32_t e3[Bls381Fq::N]; modulus_minus_one_over<Bls381Fq>(3, e3);
    const Affine<Bls381Fq> gen = group_generator<Bls381Fq>();
    Bls381Fq g = Bls381Fq::one();
    for (int tries = 0; tries < 64; tries++) {                                        // g^((p-1)/3) is a primitive cube root of unity unless g is a cube
        g = g + Bls381Fq::one();
        const Bls381Fq w = fp_pow(g, e3, Bls381Fq::N);
        if (w == Bls381Fq::one()) continue;
        FastSubgroup<Bls381Fq> c; c.beta = w;
        if (c.contains(gen)) return c;
        c.beta = w.sqr();
        if (c.contains(gen)) return c;
        break;
    }
    throw std::runtime_error("subgroup test constants: no cube root of unity makes the generator pass"); } };
#endif
// nullptr (and the [r]P path) when the group has no fast test, when CG_SUBGROUP_FULL is set, or when the constants could not be made
template <class F> const FastSubgroup<F>* fast_subgroup() {
    if (!FastSubgroup<F>::available || global_option(CG_GOPT_SUBGROUP_FULL)) return nullptr;
    static const std::pair<bool, FastSubgroup<F>> made = [] {
        try { return std::make_pair(true, FastSubgroupFactory<F>::make()); } catch (const std::exception&) { return std::make_pair(false, FastSubgroup<F>()); }
    }();
    return made.first ? &made.second : nullptr;
}
}  // namespace

struct cg_fixed_base { int curve, group; void* impl; void (*destroy)(void*); };

// ==================================================================================================== extern "C"
extern "C" {

const char* cg_last_error(void) { return g_err.c_str(); }
const char* cg_version(void) { return "cogroth16-hip 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------------------------------- bases / MSM
static int32_t bases_register_impl(cg_ctx* ctx, int32_t curve, int32_t group, const void* src, bool src_on_device, size_t n, size_t stride, int64_t inf_off, cg_bases** out) {
    if (!ctx || !out || (!src && n)) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return with_group(curve, group, [&](auto ftag, auto) -> int {
        typedef decltype(ftag) F;
        const size_t pt = sizeof(Affine<F>);
        if (stride < pt) return fail(CG_ERR_ARG, "stride smaller than a point record");
        if (inf_off >= 0 && (size_t)inf_off >= stride) return fail(CG_ERR_ARG, "infinity_offset outside the record");
        cg_bases* b = new cg_bases{ctx->device, curve, group, n, pt, nullptr};
        HIPCHK(hip_malloc_flush(&b->d_pts, std::max<size_t>(n * pt, 16)));
        if (n) {
            if (src_on_device) HIPCHK(hipMemcpyAsync(b->d_pts, src, n * pt, hipMemcpyDeviceToDevice, ctx->stream));
            else if (stride == pt && inf_off < 0) HIPCHK(hipMemcpyAsync(b->d_pts, src, n * pt, hipMemcpyHostToDevice, ctx->stream));
            else {
                void* d_raw = nullptr;
                HIPCHK(hip_malloc_flush(&d_raw, n * stride));
                HIPCHK(hipMemcpyAsync(d_raw, src, n * stride, hipMemcpyHostToDevice, ctx->stream));
                { int rc = pack_bases_launch<F>(ctx->stream, (const uint8_t*)d_raw, n, stride, (long)inf_off, (Affine<F>*)b->d_pts); if (rc) return rc; }
                HIPCHK(hipStreamSynchronize(ctx->stream));
                HIPCHK(hipFree(d_raw));
            }
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
        const int64_t compact_min_log = global_option(CG_GOPT_COMPACT_MIN_LOG);      // cg_set_option; 64 = never
        if (n >= 64 && n < ((size_t)1 << 32) && compact_min_log < 64) {   // infinity census on the packed table (registration-time work, like parsing)
            std::vector<uint8_t> host;
            const uint64_t* w = nullptr;
            if (!src_on_device && stride == pt && inf_off < 0) w = reinterpret_cast<const uint64_t*>(src);     // packed host table: scan it where it lies
            else { host.resize(n * pt); HIPCHK(hipMemcpy(host.data(), b->d_pts, n * pt, hipMemcpyDeviceToHost)); w = reinterpret_cast<const uint64_t*>(host.data()); }
            std::vector<uint32_t> live; live.reserve(n);
            const size_t words = pt / 8;
            for (size_t i = 0; i < n; i++) { uint64_t any = 0; for (size_t q = 0; q < words; q++) any |= w[i * words + q]; if (any) live.push_back((uint32_t)i); }
            b->no_inf = live.size() == n;
            // (small tables keep their records: a compacted copy gives the tables of one MSM call different scalar sets, i.e. a schedule and an
            // accumulate / reduce sequence of their own — at a few thousand points that sequence costs 0.5 ms and saves nothing.  CG_GOPT_COMPACT_MIN_LOG,
            // read per call: log2 of the smallest table that gets one)
            const size_t compact_min = (size_t)1 << std::min<int64_t>(31, std::max<int64_t>(6, compact_min_log));
            if (live.size() * 8 <= n * 7 && n >= compact_min) {
                cg_bases* cb = new cg_bases{ctx->device, curve, group, live.size(), pt, nullptr};
                cb->no_inf = true;
                HIPCHK(hip_malloc_flush(&cb->d_pts, std::max<size_t>(live.size() * pt, 16)));
                HIPCHK(hip_malloc_flush((void**)&b->d_live, std::max<size_t>(live.size() * 4, 16)));
                if (!live.empty()) {
                    HIPCHK(hipMemcpy(b->d_live, live.data(), live.size() * 4, hipMemcpyHostToDevice));
                    int rc = gather_points_launch<F>(ctx->stream, (Affine<F>*)cb->d_pts, (const Affine<F>*)b->d_pts, b->d_live, live.size()); if (rc) return rc;
                    HIPCHK(hipStreamSynchronize(ctx->stream));
                }
                uint64_t h = 1469598103934665603ull ^ (uint64_t)live.size();          // FNV-1a over the index list: equal patterns (b1 / b2) share schedules
                for (uint32_t v : live) { h ^= v; h *= 1099511628211ull; }
                b->live_sig = h ? h : 1; b->h_live = std::move(live); b->compact = cb;
            }
        }
        *out = b;
        return 0;
    });
}
int32_t cg_bases_register(cg_ctx* ctx, int32_t curve, int32_t group, const void* h_points, size_t n, size_t stride_bytes, int64_t infinity_offset, cg_bases** out) {
    return bases_register_impl(ctx, curve, group, h_points, false, n, stride_bytes, infinity_offset, out);
}
int32_t cg_bases_register_device(cg_ctx* ctx, int32_t curve, int32_t group, const void* d_points_packed, size_t n, cg_bases** out) {
    size_t pt = 0;
    int rc = with_group(curve, group, [&](auto ftag, auto) -> int { pt = sizeof(Affine<decltype(ftag)>); return 0; });
    if (rc) return rc;
    return bases_register_impl(ctx, curve, group, d_points_packed, true, n, pt, -1, out);
}
int32_t cg_bases_release(cg_bases* b) {
    if (!b) return 0;
    hipSetDevice(b->device);
    hipDeviceSynchronize();
    hipFree(b->d_pts);
    if (b->d_pre) hipFree(b->d_pre);
    if (b->d_live) hipFree(b->d_live);
    if (b->compact) { hipFree(b->compact->d_pts); if (b->compact->d_pre) hipFree(b->compact->d_pre); delete b->compact; }
    delete b;
    return 0;
}
int32_t cg_bases_check_on_curve(cg_ctx* ctx, const cg_bases* b, uint64_t* n_bad, uint64_t* first_bad) {
    if (!ctx || !b || !n_bad) return fail(CG_ERR_ARG, "null argument");
    if (b->device != ctx->device) return fail(CG_ERR_ARG, "bases live on another device");
    HIPCHK(hipSetDevice(ctx->device));
    return with_group(b->curve, b->group, [&](auto ftag, auto) -> int {
        typedef decltype(ftag) F;
        unsigned long long* d = nullptr; unsigned long long h[2] = {0ull, ~0ull};
        HIPCHK(hip_malloc_flush((void**)&d, 16));
        HIPCHK(hipMemcpyAsync(d, h, 16, hipMemcpyHostToDevice, ctx->stream));
        int rc = check_on_curve_launch<F>(ctx->stream, (const Affine<F>*)b->d_pts, b->n, CurveB<F>::get(), d);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(d));
        *n_bad = h[0]; if (first_bad) *first_bad = h[1];
        return 0;
    });
}
int32_t cg_bases_check_subgroup(cg_ctx* ctx, const cg_bases* b, uint64_t* n_bad, uint64_t* first_bad) {
    if (!ctx || !b || !n_bad) return fail(CG_ERR_ARG, "null argument");
    if (b->device != ctx->device) return fail(CG_ERR_ARG, "bases live on another device");
    *n_bad = 0; if (first_bad) *first_bad = ~0ull;
    if (b->curve == CG_BN254 && b->group == CG_G1) return 0;      // cofactor 1: every curve point is in the group
    HIPCHK(hipSetDevice(ctx->device));
    return with_group(b->curve, b->group, [&](auto ftag, auto frtag) -> int {
        typedef decltype(ftag) F; typedef decltype(frtag) Fr;
        unsigned long long* d = nullptr; unsigned long long h[2] = {0ull, ~0ull};
        HIPCHK(hip_malloc_flush((void**)&d, 16));
        HIPCHK(hipMemcpyAsync(d, h, 16, hipMemcpyHostToDevice, ctx->stream));
        const FastSubgroup<F>* fast = fast_subgroup<F>();
        int rc = fast ? check_subgroup_fast_launch<F>(ctx->stream, (const Affine<F>*)b->d_pts, b->n, *fast, d)
                      : check_subgroup_launch<F, Fr>(ctx->stream, (const Affine<F>*)b->d_pts, b->n, d);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(d));
        *n_bad = h[0]; if (first_bad) *first_bad = h[1];
        return 0;
    });
}
int32_t cg_bases_precompute(cg_ctx* ctx, cg_bases* b, int32_t c) {
    if (!ctx || !b) return fail(CG_ERR_ARG, "null argument");
    if (b->compact) return cg_bases_precompute(ctx, b->compact, c);      // MSMs only ever read the compacted copy
    // c = 0: pick by table size (measured per extra table of a shared-schedule call): 2^19 buckets only pay for themselves above
    // ~3 M points in G1 (2 M points: 5.4 ms at c = 17, 5.9 at c = 20) and above ~1.5 M in G2, whose additions cost three times as much
    // (small tables: 2^15 buckets, a shorter bit-sum reduction: 2^16-constraint step 5.25 -> 4.7 ms, 2^18 9.0 -> 8.5 ms)
    const bool auto_window = c == 0;
    if (c == 0) c = b->n > ((size_t)3 << (b->group == CG_G1 ? 20 : 19)) ? 20 : (b->n <= ((size_t)1 << 18) ? 16 : 17);
    if (c < 8 || c > 22) return fail(CG_ERR_ARG, "precompute window must be 0 (auto) or in [8, 22]");
    if (b->device != ctx->device) return fail(CG_ERR_ARG, "bases live on another device");
    if (b->n > ((size_t)1 << 24)) return fail(CG_ERR_ARG, "precomputed tables support at most 2^24 points");
    HIPCHK(hipSetDevice(ctx->device));
    if (b->d_pre) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(b->d_pre)); b->d_pre = nullptr; b->pre_c = b->pre_nwin = 0; }
    return with_group(b->curve, b->group, [&](auto ftag, auto frtag) -> int {
        typedef decltype(ftag) F; typedef decltype(frtag) Fr;
        const int nwin = Fr::Params::BITS / c + 1;
        const size_t n = std::max<size_t>(b->n, 1);
        const hipError_t e_pre = hip_malloc_flush(&b->d_pre, (size_t)nwin * n * sizeof(Affine<F>));
        if (e_pre == hipErrorOutOfMemory && auto_window) {     // the tables are an optimisation: a table that does not fit keeps the per-window bucket sets
            (void)hipGetLastError(); b->d_pre = nullptr;
            return 0;
        }
        HIPCHK(e_pre);
        Affine<F>* tab = (Affine<F>*)b->d_pre;
        HIPCHK(hipMemcpyAsync(tab, b->d_pts, b->n * sizeof(Affine<F>), hipMemcpyDeviceToDevice, ctx->stream));
        for (int j = 1; j < nwin; j++) { int rc = precompute_window_launch<F>(ctx->stream, tab + (size_t)(j - 1) * b->n, tab + (size_t)j * b->n, b->n, c); if (rc) return rc; }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        b->pre_c = c; b->pre_nwin = nwin;
        return 0;
    });
}
size_t cg_bases_len(const cg_bases* b) { return b ? b->n : 0; }

int32_t cg_bases_synth_multiples(cg_ctx* ctx, int32_t curve, int32_t group, uint64_t first, size_t n, cg_bases** out) {
    if (!ctx || !out) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return with_group(curve, group, [&](auto ftag, auto) -> int {
        typedef decltype(ftag) F;
        const uint32_t* src = curve == CG_BN254 ? (group == CG_G1 ? Bn254G1_GEN : Bn254G2_GEN) : (group == CG_G1 ? Bls381G1_GEN : Bls381G2_GEN);
        Affine<F> ga; memcpy(&ga, src, sizeof ga);
        const XYZZ<F> G = XYZZ<F>::from_affine(ga);
        int log_t = 0; while (((size_t)1 << (2 * log_t)) < n) log_t++;          // ~sqrt(n) entries per table
        const size_t T = (size_t)1 << log_t, H = std::max<size_t>(1, (n + T - 1) >> log_t);
        std::vector<XYZZ<F>> lo(T), hi(H);
        uint32_t k[2] = {(uint32_t)first, (uint32_t)(first >> 32)};
        XYZZ<F> acc = xyzz_scalar_mul(G, k, 2);
        for (size_t j = 0; j < T; j++) { lo[j] = acc; acc = xyzz_add(acc, G); }
        uint32_t kt[2] = {(uint32_t)T, (uint32_t)((uint64_t)T >> 32)};
        const XYZZ<F> step = xyzz_scalar_mul(G, kt, 2);
        acc = XYZZ<F>::infinity();
        for (size_t j = 0; j < H; j++) { hi[j] = acc; acc = xyzz_add(acc, step); }
        XYZZ<F>*d_lo = nullptr, *d_hi = nullptr;
        HIPCHK(hip_malloc_flush((void**)&d_lo, T * sizeof(XYZZ<F>))); HIPCHK(hip_malloc_flush((void**)&d_hi, H * sizeof(XYZZ<F>)));
        HIPCHK(hipMemcpy(d_lo, lo.data(), T * sizeof(XYZZ<F>), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_hi, hi.data(), H * sizeof(XYZZ<F>), hipMemcpyHostToDevice));
        cg_bases* b = new cg_bases{ctx->device, curve, group, n, sizeof(Affine<F>), nullptr};
        b->no_inf = first >= 1 && first + n > first;             // (first + i) G with 1 <= first + i < 2^64 < r is never the point at infinity
        HIPCHK(hip_malloc_flush(&b->d_pts, std::max<size_t>(n * sizeof(Affine<F>), 16)));
        int rc = synth_points_launch<F>(ctx->stream, d_lo, d_hi, log_t, n, (Affine<F>*)b->d_pts);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(d_lo)); HIPCHK(hipFree(d_hi));
        *out = b;
        return 0;
    });
}
int32_t cg_bases_from_scalars(cg_ctx* ctx, int32_t curve, int32_t group, const void* d_scalars, size_t n, cg_bases** out) {
    if (!ctx || !out || (!d_scalars && n)) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return with_group(curve, group, [&](auto ftag, auto frtag) -> int {
        typedef decltype(ftag) F; typedef decltype(frtag) Fr;
        const uint32_t* src = curve == CG_BN254 ? (group == CG_G1 ? Bn254G1_GEN : Bn254G2_GEN) : (group == CG_G1 ? Bls381G1_GEN : Bls381G2_GEN);
        Affine<F> ga; memcpy(&ga, src, sizeof ga);
        const int nwin = (Fr::Params::BITS + 7) / 8;
        Affine<F>* d_tab = nullptr;
        HIPCHK(hip_malloc_flush((void**)&d_tab, (size_t)nwin * 255 * sizeof(Affine<F>)));
        cg_bases* b = new cg_bases{ctx->device, curve, group, n, sizeof(Affine<F>), nullptr};
        hipError_t e = hip_malloc_flush(&b->d_pts, std::max<size_t>(n * sizeof(Affine<F>), 16));
        if (e != hipSuccess) { hipFree(d_tab); delete b; return fail(CG_ERR_OOM, "cg_bases_from_scalars: out of device memory"); }
        int rc = fixed_base_mul_launch<F, Fr>(ctx->stream, ga, (const Fr*)d_scalars, n, d_tab, (Affine<F>*)b->d_pts);
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        hipFree(d_tab);
        if (rc || e2 != hipSuccess) { hipFree(b->d_pts); delete b; return rc ? rc : fail(CG_ERR_HIP, std::string("cg_bases_from_scalars: ") + hipGetErrorString(e2)); }
        *out = b;
        return 0;
    });
}
int32_t cg_bases_download(cg_ctx* ctx, const cg_bases* bases, size_t offset, size_t n, void* h_out_packed) {
    if (!ctx || !bases || !h_out_packed) return fail(CG_ERR_ARG, "null argument");
    if (offset + n > bases->n) return fail(CG_ERR_ARG, "slice out of range");
    HIPCHK(hipMemcpyAsync(h_out_packed, (const char*)bases->d_pts + offset * bases->pt_bytes, n * bases->pt_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int32_t cg_ctx_set_option(cg_ctx* ctx, int32_t option, int64_t value) {
    if (!ctx) return fail(CG_ERR_ARG, "null ctx");
    switch (option) {
        case CG_OPT_MSM_CHUNK: return cg_msm_set_chunk(ctx, (int32_t)value);
        case CG_OPT_MSM_WINDOW: return cg_msm_set_window(ctx, (int32_t)value);
        case CG_OPT_MSM_SCATTER_CAP: return cg_msm_set_scatter_capacity(ctx, (int32_t)value);
        case CG_OPT_MSM_TABLE_ORDER: if (value < 0 || value > 2) break; ctx->table_order = (int)value; return 0;
        case CG_OPT_MSM_G2_AFTER: if (value < -1 || value > 64) break; ctx->g2_after = (int)value; return 0;
        case CG_OPT_MSM_G2_SLICES: if (value < 0 || value > 1) break; ctx->g2_slices = (int)value; return 0;
        case CG_OPT_MSM_REDUCE_BATCH: if (value < 0 || value > 3) break; ctx->red_batch = (int)value; return 0;
        case CG_OPT_MSM_ACC_SLOTS: if (value < 2 || value > cg_ctx::ACC_SLOTS_MAX) break; ctx->acc_slots = (int)value; return 0;
        case CG_OPT_MSM_WIDE_SMALL: if (value < 0 || value > 30 || (value > 1 && value < 10)) break; ctx->wide_small = (int)value; return 0;
        case CG_OPT_MSM_ONE_STREAM_LOG: if (value < 0 || value > 30) break; ctx->one_stream_log = (int)value; return 0;
        case CG_OPT_MSM_OFF_MAIN_LOG: if (value < 0 || value > 30) break; ctx->off_main_log = (int)value; return 0;
        case CG_OPT_MSM_SOLO_LOG: if (value < 0 || value > 30) break; ctx->solo_log = (int)value; return 0;
        default: return fail(CG_ERR_ARG, "cg_ctx_set_option: unknown option");
    }
    return fail(CG_ERR_ARG, "cg_ctx_set_option: value out of range");
}
int32_t cg_ctx_get_option(const cg_ctx* ctx, int32_t option, int64_t* value) {
    if (!ctx || !value) return fail(CG_ERR_ARG, "null argument");
    switch (option) {
        case CG_OPT_MSM_CHUNK: *value = ctx->msm_chunk; return 0;
        case CG_OPT_MSM_WINDOW: *value = ctx->msm_window; return 0;
        case CG_OPT_MSM_SCATTER_CAP: *value = ctx->scatter_cap; return 0;
        case CG_OPT_MSM_TABLE_ORDER: *value = ctx->table_order; return 0;
        case CG_OPT_MSM_G2_AFTER: *value = ctx->g2_after; return 0;
        case CG_OPT_MSM_G2_SLICES: *value = ctx->g2_slices; return 0;
        case CG_OPT_MSM_REDUCE_BATCH: *value = ctx->red_batch; return 0;
        case CG_OPT_MSM_ACC_SLOTS: *value = ctx->acc_slots; return 0;
        case CG_OPT_MSM_WIDE_SMALL: *value = ctx->wide_small; return 0;
        case CG_OPT_MSM_ONE_STREAM_LOG: *value = ctx->one_stream_log; return 0;
        case CG_OPT_MSM_OFF_MAIN_LOG: *value = ctx->off_main_log; return 0;
        case CG_OPT_MSM_SOLO_LOG: *value = ctx->solo_log; return 0;
        default: return fail(CG_ERR_ARG, "cg_ctx_get_option: unknown option");
    }
}
int32_t cg_set_option(int32_t option, int64_t value) {
    if (option < 1 || option >= CG_GOPT_COUNT) return fail(CG_ERR_ARG, "cg_set_option: unknown option");
    if (value < 0 || (option != CG_GOPT_COMPACT_MIN_LOG && value > 1) || value > 64) return fail(CG_ERR_ARG, "cg_set_option: value out of range");
    g_options.v[option].store(value); return 0;
}
int32_t cg_get_option(int32_t option, int64_t* value) {
    if (option < 1 || option >= CG_GOPT_COUNT || !value) return fail(CG_ERR_ARG, "cg_get_option: unknown option");
    *value = g_options.v[option].load(); return 0;
}
// ---------------------------------------------------------------------------------------------------- vector ops
int32_t cg_vec_add_dev(cg_ctx* ctx, int32_t curve, void* o, const void* a, const void* b, size_t n) { return vec_binary<0>(ctx, curve, o, a, b, n); }
int32_t cg_vec_sub_dev(cg_ctx* ctx, int32_t curve, void* o, const void* a, const void* b, size_t n) { return vec_binary<1>(ctx, curve, o, a, b, n); }
int32_t cg_vec_mul_dev(cg_ctx* ctx, int32_t curve, void* o, const void* a, const void* b, size_t n) { return vec_binary<2>(ctx, curve, o, a, b, n); }

int32_t cg_vec_rep3_mul_local_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_aa, const void* d_ab, const void* d_ba, const void* d_bb, const void* d_mask, size_t n) {
    if (!ctx || !d_out || !d_aa || !d_ab || !d_ba || !d_bb) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        StatScope ss(ctx, TAG_VEC);
        return launch_rep3_mul_local<Fr>(ctx->stream, (Fr*)d_out, (const Fr*)d_aa, (const Fr*)d_ab, (const Fr*)d_ba, (const Fr*)d_bb, (const Fr*)d_mask, n);
    });
}
// one attempt of n draws with `margin` surplus draws' worth of candidates: kernels + the count's download enqueued, nothing waited for
static int rand_draw_begin(cg_ctx* ctx, int32_t curve, const uint8_t* seed32, uint64_t word_pos, size_t n, void* d_out, double margin, int* slot_out) {
    int slot = -1;
    for (int i = 0; i < cg_ctx::RAND_DRAWS; i++) if (!ctx->rand_draw[i].live) { slot = i; break; }
    if (slot < 0) return fail(CG_ERR_ARG, "cg_chacha12_fr_rand_dev_begin: too many draws in flight (finish one first)");
    if (!ctx->rand_result) HIPCHK(hipHostMalloc((void**)&ctx->rand_result, sizeof(unsigned long long) * 2 * cg_ctx::RAND_DRAWS, hipHostMallocDefault));
    cg_ctx::RandDraw& d = ctx->rand_draw[slot];
    if (!d.ev) HIPCHK(hipEventCreateWithFlags(&d.ev, hipEventDisableTiming));
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        typedef typename Fr::Params P;
        StatScope ss(ctx, TAG_VEC);
        uint32_t key[8];
        for (int i = 0; i < 8; i++) key[i] = (uint32_t)seed32[4 * i] | (uint32_t)seed32[4 * i + 1] << 8 | (uint32_t)seed32[4 * i + 2] << 16 | (uint32_t)seed32[4 * i + 3] << 24;
        // acceptance rate = modulus / 2^BITS (BN254 Fr 0.756, BLS12-381 Fr 0.906)
        const double accept = (double)P::P[7] / (double)(1ull << (P::BITS - 224));
        const uint64_t n_cand = (uint64_t)(((double)n + margin) / accept) + 2;
        const uint64_t n_pairs = n_cand / 2 + 2;
        const uint64_t tiles = (n_pairs + 255) / 256;
        d.d_cand = d.d_small = nullptr;
        if (int rc = cg_dev_alloc(ctx, n_pairs * 64, &d.d_cand)) return rc;
        if (int rc = cg_dev_alloc(ctx, tiles * 4 + 16, &d.d_small)) { cg_dev_free(ctx, d.d_cand); return rc; }
        unsigned long long* h = ctx->rand_result + 2 * slot; h[0] = h[1] = 0;
        int rc = chacha12_fr_rand_launch(ctx->stream, key, P::P, P::BITS, word_pos, n_pairs, n, d.d_cand, (uint32_t*)((char*)d.d_small + 16), (unsigned long long*)d.d_small, d_out);
        hipError_t e = rc ? hipSuccess : hipMemcpyAsync(h, d.d_small, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
        if (!rc && e == hipSuccess) e = hipEventRecord(d.ev, ctx->stream);
        if (rc || e != hipSuccess) { hipStreamSynchronize(ctx->stream); cg_dev_free(ctx, d.d_cand); cg_dev_free(ctx, d.d_small); if (rc) return rc; HIPCHK(e); }
        d.live = true; d.word_pos = word_pos; d.n = n;
        *slot_out = slot;
        return 0;
    });
}
// waits for the attempt; *enough = the n-th accepted candidate was among those generated
static int rand_draw_finish(cg_ctx* ctx, int slot, bool* enough, uint64_t* word_pos_after) {
    cg_ctx::RandDraw& d = ctx->rand_draw[slot];
    hipError_t e = hipEventSynchronize(d.ev);
    { void* two[2] = {d.d_cand, d.d_small}; cg_dev_free_many(ctx, two, 2); }
    d.live = false;
    HIPCHK(e);
    const unsigned long long* h = ctx->rand_result + 2 * slot;
    *enough = h[0] >= d.n;
    if (*enough && word_pos_after) *word_pos_after = d.word_pos + 8 * ((uint64_t)h[1] + 1);
    return 0;
}
static int rand_draw_args(cg_ctx* ctx, const uint8_t* seed32, uint64_t word_pos, size_t n, void* d_out, const char* who) {
    if (!ctx || !seed32 || (n && !d_out)) return fail(CG_ERR_ARG, "null argument");
    if (n >= ((size_t)1 << 31) || word_pos > (~0ull >> 1)) return fail(CG_ERR_ARG, std::string(who) + ": size or position out of range");
    return 0;
}
int32_t cg_chacha12_fr_rand_dev(cg_ctx* ctx, int32_t curve, const uint8_t* seed32, uint64_t word_pos, size_t n, void* d_out, uint64_t* word_pos_after) {
    if (int rc = rand_draw_args(ctx, seed32, word_pos, n, d_out, "cg_chacha12_fr_rand_dev")) return rc;
    if (n == 0) { if (word_pos_after) *word_pos_after = word_pos; return 0; }
    HIPCHK(hipSetDevice(ctx->device));
    double margin = 8.0 * std::sqrt((double)n) + 64.0;           // candidates for n draws + 8 standard deviations + a floor; four times as many after a shortfall
    for (int attempt = 0; attempt < 4; attempt++, margin *= 4.0) {
        int slot = -1; bool enough = false;
        if (int rc = rand_draw_begin(ctx, curve, seed32, word_pos, n, d_out, margin, &slot)) return rc;
        if (int rc = rand_draw_finish(ctx, slot, &enough, word_pos_after)) return rc;
        if (enough) return 0;
    }
    return fail(CG_ERR_HIP, "cg_chacha12_fr_rand_dev: too few accepted candidates");
}
int32_t cg_chacha12_fr_rand_dev_begin(cg_ctx* ctx, int32_t curve, const uint8_t* seed32, uint64_t word_pos, size_t n, void* d_out, int32_t* ticket) {
    if (!ticket) return fail(CG_ERR_ARG, "null argument");
    if (int rc = rand_draw_args(ctx, seed32, word_pos, n, d_out, "cg_chacha12_fr_rand_dev_begin")) return rc;
    if (n == 0) return fail(CG_ERR_ARG, "cg_chacha12_fr_rand_dev_begin: nothing to draw");
    HIPCHK(hipSetDevice(ctx->device));
    int slot = -1;
    // no second attempt is possible once the consumers of d_out are enqueued: 12 standard deviations of surplus (a shortfall every ~10^32 calls)
    if (int rc = rand_draw_begin(ctx, curve, seed32, word_pos, n, d_out, 12.0 * std::sqrt((double)n) + 64.0, &slot)) return rc;
    *ticket = slot;
    return 0;
}
int32_t cg_chacha12_fr_rand_dev_finish(cg_ctx* ctx, int32_t ticket, uint64_t* word_pos_after) {
    if (!ctx) return fail(CG_ERR_ARG, "null argument");
    if (ticket < 0 || ticket >= cg_ctx::RAND_DRAWS || !ctx->rand_draw[ticket].live) return fail(CG_ERR_ARG, "cg_chacha12_fr_rand_dev_finish: no such draw in flight");
    HIPCHK(hipSetDevice(ctx->device));
    bool enough = false;
    if (int rc = rand_draw_finish(ctx, ticket, &enough, word_pos_after)) return rc;
    if (!enough) return fail(CG_ERR_HIP, "cg_chacha12_fr_rand_dev_finish: too few accepted candidates (d_out is incomplete: draw again with cg_chacha12_fr_rand_dev)");
    return 0;
}
int32_t cg_vec_check_canonical_dev(cg_ctx* ctx, int32_t curve, const void* d_vec, size_t n, void* d_count) {
    if (!ctx || !d_vec || !d_count) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        StatScope ss(ctx, TAG_VEC);
        return launch_vec_count_noncanonical<Fr>(ctx->stream, (const Fr*)d_vec, n, (unsigned long long*)d_count);
    });
}
int32_t cg_vec_affine_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_a, size_t n, const void* h_c, const void* h_d) {
    if (!ctx || !d_out || !d_a || !h_c) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        Fr c, d = Fr::zero(); copy_in(c, h_c); if (h_d) copy_in(d, h_d);
        StatScope ss(ctx, TAG_VEC);
        return launch_vec_affine<Fr>(ctx->stream, (Fr*)d_out, (const Fr*)d_a, n, c, d);
    });
}
int32_t cg_vec_fill_dev(cg_ctx* ctx, int32_t curve, void* d_v, size_t n, const void* h_value) {
    if (!ctx || !d_v || !h_value) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        Fr v; copy_in(v, h_value);
        StatScope ss(ctx, TAG_VEC);
        return launch_vec_fill<Fr>(ctx->stream, (Fr*)d_v, n, v);
    });
}
int32_t cg_vec_gather_strided_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_in, size_t n, size_t offset, size_t stride) {
    if (!ctx || !d_out || !d_in) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        StatScope ss(ctx, TAG_VEC);
        return launch_vec_gather_strided<Fr>(ctx->stream, (Fr*)d_out, (const Fr*)d_in, n, offset, stride);
    });
}
int32_t cg_vec_lincomb_dev(cg_ctx* ctx, int32_t curve, void* d_out, int64_t out_off, int64_t out_stride, size_t n, int32_t n_terms,
                           const void* const* d_src, const int64_t* src_off, const int64_t* src_stride, const void* h_coeffs) {
    if (!ctx || !d_out || !d_src || !src_off || !src_stride || !h_coeffs) return fail(CG_ERR_ARG, "null argument");
    if (n_terms < 1 || n_terms > LINCOMB_MAX) return fail(CG_ERR_ARG, "cg_vec_lincomb_dev: 1..8 terms");
    HIPCHK(hipSetDevice(ctx->device));
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        LincombArgs<Fr> a; memset(&a, 0, sizeof a);
        a.n_terms = n_terms;
        const Fr one = Fr::one();
        for (int j = 0; j < n_terms; j++) {
            if (!d_src[j]) return fail(CG_ERR_ARG, "null source vector");
            a.src[j] = (const Fr*)d_src[j]; a.off[j] = src_off[j]; a.stride[j] = src_stride[j];
            copy_in(a.coeff[j], (const char*)h_coeffs + (size_t)j * sizeof(Fr));
            a.unit[j] = memcmp(&a.coeff[j], &one, sizeof(Fr)) == 0;
        }
        StatScope ss(ctx, TAG_VEC);
        return launch_vec_lincomb<Fr>(ctx->stream, (Fr*)d_out, out_off, out_stride, n, a);
    });
}
static int32_t prefix_scan_dev(cg_ctx* ctx, int32_t curve, int op, void* d_out, const void* d_in, size_t n) {
    if (!ctx || !d_out || !d_in) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return 0;
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        const size_t ntiles = (n + 2047) / 2048;
        { int rc = ensure_arena(ctx, align_up(ntiles * sizeof(Fr))); if (rc) return rc; }
        StatScope ss(ctx, TAG_VEC);
        return launch_prefix_scan<Fr>(ctx->stream, op, (Fr*)d_out, (const Fr*)d_in, n, (Fr*)ctx->arena.base);
    });
}
int32_t cg_vec_prefix_prod_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_in, size_t n) { return prefix_scan_dev(ctx, curve, 0, d_out, d_in, n); }
int32_t cg_vec_prefix_sum_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_in, size_t n) { return prefix_scan_dev(ctx, curve, 1, d_out, d_in, n); }
int32_t cg_vec_inverse_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_in, size_t n) {
    if (!ctx || !d_out || !d_in) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        StatScope ss(ctx, TAG_VEC);
        return launch_vec_inverse<Fr>(ctx->stream, (Fr*)d_out, (const Fr*)d_in, n);
    });
}
int32_t cg_spmv_csr_dev(cg_ctx* ctx, int32_t curve, const uint32_t* d_row_ptr, const uint32_t* d_col, const void* d_coeff, size_t n_rows,
                        const void* d_pub, uint32_t n_inputs, int32_t party, const void* d_wit_a, const void* d_wit_b, void* d_out_a, void* d_out_b) {
    if (!ctx || !d_row_ptr || !d_out_a || !d_wit_a) return fail(CG_ERR_ARG, "null argument");
    if (party < -1 || party > 2) return fail(CG_ERR_ARG, "party must be -1 (single component) or 0..2");
    if (party >= 0 && (!d_wit_b || !d_out_b)) return fail(CG_ERR_ARG, "REP3 needs both share components");
    HIPCHK(hipSetDevice(ctx->device));
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        StatScope ss(ctx, TAG_SPMV);
        return launch_spmv_csr<Fr>(ctx->stream, d_row_ptr, d_col, (const Fr*)d_coeff, n_rows, (const Fr*)d_pub, n_inputs, (int)party,
                                   (const Fr*)d_wit_a, (const Fr*)d_wit_b, (Fr*)d_out_a, (Fr*)d_out_b);
    });
}

static int32_t host_vec_call(cg_ctx* ctx, size_t n, int n_in, const void* const* h_in, void* h_out, int (*fn)(cg_ctx*, void* const*, void*, size_t, int), int curve) {
    if (!ctx || !h_out) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<void*> d(n_in + 1, nullptr);
    const size_t bytes = std::max<size_t>(n * 32, 16);
    for (int j = 0; j <= n_in; j++) HIPCHK(hip_malloc_flush(&d[j], bytes));
    for (int j = 0; j < n_in; j++) if (h_in[j]) HIPCHK(hipMemcpyAsync(d[j], h_in[j], n * 32, hipMemcpyHostToDevice, ctx->stream));
    std::vector<void*> args(d.begin(), d.begin() + n_in);
    for (int j = 0; j < n_in; j++) if (!h_in[j]) args[j] = nullptr;
    int rc = fn(ctx, args.data(), d[n_in], n, curve);
    if (!rc) { hipError_t e = hipMemcpyAsync(h_out, d[n_in], n * 32, hipMemcpyDeviceToHost, ctx->stream); if (e != hipSuccess) rc = fail(CG_ERR_HIP, hipGetErrorString(e)); }
    hipStreamSynchronize(ctx->stream);
    for (auto p : d) hipFree(p);
    return rc;
}
int32_t cg_vec_mul(cg_ctx* ctx, int32_t curve, void* h_out, const void* h_a, const void* h_b, size_t n) {
    const void* in[2] = {h_a, h_b};
    return host_vec_call(ctx, n, 2, in, h_out, [](cg_ctx* c, void* const* a, void* o, size_t n, int curve) { return (int)cg_vec_mul_dev(c, curve, o, a[0], a[1], n); }, curve);
}
int32_t cg_vec_rep3_mul_local(cg_ctx* ctx, int32_t curve, void* h_out, const void* h_aa, const void* h_ab, const void* h_ba, const void* h_bb, const void* h_mask, size_t n) {
    const void* in[5] = {h_aa, h_ab, h_ba, h_bb, h_mask};
    return host_vec_call(ctx, n, 5, in, h_out, [](cg_ctx* c, void* const* a, void* o, size_t n, int curve) { return (int)cg_vec_rep3_mul_local_dev(c, curve, o, a[0], a[1], a[2], a[3], a[4], n); }, curve);
}

// ---------------------------------------------------------------------------------------------------- O(1) host helpers
// (all on 64-bit limbs, host_ec64.hpp: a small proof's tail is a few dozen of these calls and nothing else — the additions, the seven
// conversions to affine form (an inversion each) and the subgroup test of the one G2 point a REP3 party receives were 0.25 ms of a 1.1 ms
// proof on the 32-bit-limb field code the kernels share with the host)
int32_t cg_point_add(int32_t curve, int32_t group, const void* h_a, const void* h_b, void* h_out) {
    if (!h_a || !h_b || !h_out) return fail(CG_ERR_ARG, "null argument");
    return with_group64(curve, group, [&](auto ftag, auto) -> int {
        typedef decltype(ftag) F;
        cg64::Jac<F> a, b; memcpy(&a, h_a, sizeof a); memcpy(&b, h_b, sizeof b);
        const cg64::Jac<F> r = cg64::add(a, b);
        memcpy(h_out, &r, sizeof r); return 0;
    });
}
int32_t cg_point_neg(int32_t curve, int32_t group, const void* h_a, void* h_out) {
    return with_group(curve, group, [&](auto ftag, auto) -> int {
        typedef decltype(ftag) F;
        Jacobian<F> a; memcpy(&a, h_a, sizeof a); a.y = a.y.neg(); memcpy(h_out, &a, sizeof a); return 0;
    });
}
int32_t cg_point_scalar_mul(int32_t curve, int32_t group, const void* h_a, const void* h_k, void* h_out) {
    if (!h_a || !h_k || !h_out) return fail(CG_ERR_ARG, "null argument");
    return with_group64(curve, group, [&](auto ftag, auto frtag) -> int {     // 64-bit limbs, 4-bit windows (host_ec64.hpp)
        typedef decltype(ftag) F; typedef decltype(frtag) Fr;
        cg64::Jac<F> a; static_assert(sizeof a == 3 * sizeof(F), ""); memcpy(&a, h_a, sizeof a);
        Fr k; memcpy(k.v, h_k, sizeof k.v); k = k.from_mont();
        // a scalar just below the group order is a small negative number (the Lagrange coefficients of Shamir's openings: -1, -2, -3): multiply
        // by its negation — a handful of window steps instead of 64 — and negate the point
        const Fr kn = k.neg();                                                         // (limbs are canonical either way: from_mont reduces)
        bool small_neg = !kn.is_zero(); for (int i = 1; i < Fr::N; i++) small_neg = small_neg && kn.v[i] == 0;
        cg64::Jac<F> r = cg64::scalar_mul(a, small_neg ? kn.v : k.v, Fr::N);
        if (small_neg) r = cg64::neg(r);
        memcpy(h_out, &r, sizeof r); return 0;
    });
}
// Fixed-base tables for the points a session multiplies in every proof (delta_1, delta_2, the generators, the public-input records of
// the a / b1 / b2 queries): 8-bit windows, one mixed addition per scalar byte (~10 us for G1 against ~60 us variable-base).
int32_t cg_fixed_base_create(int32_t curve, int32_t group, const void* h_point_jacobian, cg_fixed_base** out) {
    if (!h_point_jacobian || !out) return fail(CG_ERR_ARG, "null argument");
    return with_group64(curve, group, [&](auto ftag, auto frtag) -> int {
        typedef decltype(ftag) F; typedef decltype(frtag) Fr;
        cg64::Jac<F> a; memcpy(&a, h_point_jacobian, sizeof a);
        auto* fb = new cg64::FixedBase<F>();
        fb->build(a, Fr::N);
        *out = new cg_fixed_base{curve, group, fb, [](void* p) { delete (cg64::FixedBase<F>*)p; }};
        return 0;
    });
}
int32_t cg_fixed_base_mul(const cg_fixed_base* t, const void* h_k, void* h_out_jacobian) {
    if (!t || !h_k || !h_out_jacobian) return fail(CG_ERR_ARG, "null argument");
    return with_group64(t->curve, t->group, [&](auto ftag, auto frtag) -> int {
        typedef decltype(ftag) F; typedef decltype(frtag) Fr;
        Fr k; memcpy(k.v, h_k, sizeof k.v); k = k.from_mont();
        const cg64::Jac<F> r = ((const cg64::FixedBase<F>*)t->impl)->mul(k.v);
        memcpy(h_out_jacobian, &r, sizeof r); return 0;
    });
}
int32_t cg_fixed_base_destroy(cg_fixed_base* t) { if (t) { t->destroy(t->impl); delete t; } return 0; }
int32_t cg_point_to_affine(int32_t curve, int32_t group, const void* h_a, void* h_out_affine) {
    if (!h_a || !h_out_affine) return fail(CG_ERR_ARG, "null argument");
    return with_group64(curve, group, [&](auto ftag, auto) -> int {
        typedef decltype(ftag) F;
        cg64::Jac<F> a; memcpy(&a, h_a, sizeof a);
        const cg64::Aff<F> r = cg64::to_affine(a);
        memcpy(h_out_affine, &r, sizeof r); return 0;
    });
}
int32_t cg_point_from_affine(int32_t curve, int32_t group, const void* h_affine, void* h_out) {
    return with_group(curve, group, [&](auto ftag, auto) -> int {
        typedef decltype(ftag) F;
        Affine<F> a; memcpy(&a, h_affine, sizeof a);
        Jacobian<F> r = xyzz_to_jacobian(XYZZ<F>::from_affine(a));
        memcpy(h_out, &r, sizeof r); return 0;
    });
}
// the checks a deserialised point gets in the reference (ark-serialize with Validate::Yes, as mpc-net's receivers use it): coordinates
// below the modulus, on the curve, in the prime-order subgroup.  Host arithmetic: for the handful of points a proof receives from its peers.
extern "C++" {
namespace {
template <class P> bool limbs_below_modulus(const cg::Fp<P>& a) {
    for (int i = P::N - 1; i >= 0; i--) { if (a.v[i] < P::P[i]) return true; if (a.v[i] > P::P[i]) return false; }
    return false;
}
template <class B> bool limbs_below_modulus(const cg::Fp2<B>& a) { return limbs_below_modulus(a.c0) && limbs_below_modulus(a.c1); }
// the 64-bit-limb twin of a coordinate field, and a value carried over (both are little-endian Montgomery forms with the same R: the same bytes)
template <class F32> struct Host64;
template <> struct Host64<Bn254Fq> { typedef H64BnFq type; };
template <> struct Host64<cg::Fp2<Bn254Fq>> { typedef cg64::Fp2<H64BnFq> type; };
#if CG_WITH_BLS
template <> struct Host64<Bls381Fq> { typedef H64BlsFq type; };
template <> struct Host64<cg::Fp2<Bls381Fq>> { typedef cg64::Fp2<H64BlsFq> type; };
#endif
template <class F32> typename Host64<F32>::type as64(const F32& v) { typename Host64<F32>::type r; static_assert(sizeof r == sizeof v, "limb forms differ in size"); memcpy(&r, &v, sizeof r); return r; }
template <class B64> cg64::Jac<cg64::Fp2<B64>> psi64(const cg64::Jac<cg64::Fp2<B64>>& p, const cg64::Fp2<B64>& gx, const cg64::Fp2<B64>& gy) {
    if (p.is_inf()) return p;
    return {p.x.conj() * gx, p.y.conj() * gy, p.z.conj()};                             // (conj(X) gx, conj(Y) gy, conj(Z)): the affine map on x = X / Z^2, y = Y / Z^3
}
// the endomorphism tests of subgroup.hpp (FastSubgroup<F>::contains) with the same constants, on 64-bit limbs
bool subgroup64(const FastSubgroup<cg::Fp2<Bn254Fq>>& t, const cg64::Aff<cg64::Fp2<H64BnFq>>& p) {
    const auto gx = as64(t.psi.gx), gy = as64(t.psi.gy);
    auto e = cg64::mul_u64(p, FastSubgroup<cg::Fp2<Bn254Fq>>::X);                       // [x]P
    auto lhs = cg64::madd(e, p);                                                       // [x + 1]P
    e = psi64(e, gx, gy); lhs = cg64::add(lhs, e);
    e = psi64(e, gx, gy); lhs = cg64::add(lhs, e);
    e = psi64(e, gx, gy);
    return cg64::same_point(lhs, cg64::dbl(e));
}
#if CG_WITH_BLS
bool subgroup64(const FastSubgroup<cg::Fp2<Bls381Fq>>& t, const cg64::Aff<cg64::Fp2<H64BlsFq>>& p) {
    typedef cg64::Fp2<H64BlsFq> F;
    const cg64::Jac<F> q = cg64::mul_u64(p, FastSubgroup<cg::Fp2<Bls381Fq>>::X_ABS);
    return cg64::same_point(psi64(cg64::Jac<F>{p.x, p.y, F::one()}, as64(t.psi.gx), as64(t.psi.gy)), cg64::neg(q));
}
bool subgroup64(const FastSubgroup<Bls381Fq>& t, const cg64::Aff<H64BlsFq>& p) {
    typedef H64BlsFq F;
    const cg64::Jac<F> q = cg64::mul_u64(cg64::mul_u64(p, FastSubgroup<Bls381Fq>::X_ABS), FastSubgroup<Bls381Fq>::X_ABS);   // [x^2]P
    return cg64::same_point(cg64::Jac<F>{p.x * as64(t.beta), p.y, F::one()}, cg64::neg(q));
}
#endif
template <class F32, class A64> bool subgroup64(const FastSubgroup<F32>&, const A64&) { return true; }   // (groups without a fast test never get here)
}
}  // extern "C++"
int32_t cg_point_validate(int32_t curve, int32_t group, const void* h_affine, int32_t* ok) {
    if (!h_affine || !ok) return fail(CG_ERR_ARG, "null argument");
    return with_group(curve, group, [&](auto ftag, auto frtag) -> int {
        typedef decltype(ftag) F; typedef decltype(frtag) Fr;
        Affine<F> a; memcpy(&a, h_affine, sizeof a);
        *ok = 0;
        if (!limbs_below_modulus(a.x) || !limbs_below_modulus(a.y)) return 0;
        if (a.is_inf()) { *ok = 1; return 0; }
        typedef typename Host64<F>::type F64;
        const cg64::Aff<F64> a64{as64(a.x), as64(a.y)};
        if (!(a64.y.sqr() == a64.x.sqr() * a64.x + as64(CurveB<F>::get()))) return 0;
        if (const FastSubgroup<F>* fast = fast_subgroup<F>()) { *ok = subgroup64(*fast, a64) ? 1 : 0; return 0; }
        if (!(curve == CG_BN254 && group == CG_G1)) {                                    // cofactor 1 there
            XYZZ<F> r = XYZZ<F>::infinity();
            for (int b = Fr::Params::BITS - 1; b >= 0; b--) {
                r = xyzz_dbl(r);
                if ((Fr::Params::P[b >> 5] >> (b & 31)) & 1u) r = xyzz_madd(r, a.x, a.y);
            }
            if (!r.is_inf()) return 0;
        }
        *ok = 1; return 0;
    });
}
int32_t cg_fr_is_canonical(int32_t curve, const void* h_in, size_t n, int32_t* ok) {
    if (!h_in || !ok) return fail(CG_ERR_ARG, "null argument");
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        *ok = 1;
        for (size_t i = 0; i < n && *ok; i++) { Fr a; memcpy(a.v, (const uint8_t*)h_in + i * sizeof a.v, sizeof a.v); if (!limbs_below_modulus(a)) *ok = 0; }
        return 0;
    });
}
int32_t cg_fr_op(int32_t curve, int32_t op, const void* h_a, const void* h_b, void* h_out) {
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        Fr a, b = Fr::zero(); copy_in(a, h_a); if (h_b) copy_in(b, h_b);
        Fr r;
        switch (op) { case 0: r = a + b; break; case 1: r = a - b; break; case 2: r = a * b; break; case 3: r = fp_inverse(a); break; default: return fail(CG_ERR_ARG, "bad op"); }
        memcpy(h_out, r.v, sizeof r.v); return 0;
    });
}

int32_t cg_fr_from_canonical(int32_t curve, const void* h_in, void* h_out, size_t n) {
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        const uint8_t* in = (const uint8_t*)h_in; uint8_t* out = (uint8_t*)h_out;
        for (size_t i = 0; i < n; i++) {
            Fr a; memcpy(a.v, in + i * sizeof a.v, sizeof a.v);
            // reduce: top limb of r has >= 2 spare... subtract r while a >= r (at most 7 times for 256-bit inputs)
            for (int it = 0; it < 8; it++) {
                uint32_t d[Fr::N]; uint32_t borrow = 0;
                for (int l = 0; l < Fr::N; l++) { uint64_t x = (uint64_t)a.v[l] - Fr::Params::P[l] - borrow; d[l] = (uint32_t)x; borrow = (uint32_t)(x >> 32) & 1u; }
                if (borrow) break;
                for (int l = 0; l < Fr::N; l++) a.v[l] = d[l];
            }
            Fr m = a.to_mont();
            memcpy(out + i * sizeof a.v, m.v, sizeof m.v);
        }
        return 0;
    });
}
int32_t cg_fr_to_canonical(int32_t curve, const void* h_in, void* h_out, size_t n) {
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        const uint8_t* in = (const uint8_t*)h_in; uint8_t* out = (uint8_t*)h_out;
        for (size_t i = 0; i < n; i++) { Fr a; memcpy(a.v, in + i * sizeof a.v, sizeof a.v); Fr c = a.from_mont(); memcpy(out + i * sizeof a.v, c.v, sizeof c.v); }
        return 0;
    });
}
// base-field coordinates <-> canonical little-endian (proof / verification-key JSON carries decimal canonical values, traits.rs:186-233)
int32_t cg_fq_to_canonical(int32_t curve, const void* h_in, void* h_out, size_t n) {
    return with_fq(curve, [&](auto ftag) -> int {
        typedef decltype(ftag) Fq;
        const uint8_t* in = (const uint8_t*)h_in; uint8_t* out = (uint8_t*)h_out;
        for (size_t i = 0; i < n; i++) { Fq a; memcpy(a.v, in + i * sizeof a.v, sizeof a.v); Fq c = a.from_mont(); memcpy(out + i * sizeof a.v, c.v, sizeof c.v); }
        return 0;
    });
}
int32_t cg_fq_from_canonical(int32_t curve, const void* h_in, void* h_out, size_t n) {   // input must be < q
    return with_fq(curve, [&](auto ftag) -> int {
        typedef decltype(ftag) Fq;
        const uint8_t* in = (const uint8_t*)h_in; uint8_t* out = (uint8_t*)h_out;
        for (size_t i = 0; i < n; i++) {
            Fq a; memcpy(a.v, in + i * sizeof a.v, sizeof a.v);
            for (int l = Fq::N - 1; l >= 0; l--) { if (a.v[l] < Fq::Params::P[l]) break; if (a.v[l] > Fq::Params::P[l] || l == 0) return fail(CG_ERR_ARG, "coordinate not reduced"); }
            Fq m = a.to_mont(); memcpy(out + i * sizeof a.v, m.v, sizeof m.v);
        }
        return 0;
    });
}
int32_t cg_point_generator(int32_t curve, int32_t group, void* h_out) {
    return with_group(curve, group, [&](auto ftag, auto) -> int {
        typedef decltype(ftag) F;
        const uint32_t* src = curve == CG_BN254 ? (group == CG_G1 ? Bn254G1_GEN : Bn254G2_GEN) : (group == CG_G1 ? Bls381G1_GEN : Bls381G2_GEN);
        Affine<F> a; memcpy(&a, src, sizeof a);
        Jacobian<F> r = xyzz_to_jacobian(XYZZ<F>::from_affine(a));
        memcpy(h_out, &r, sizeof r); return 0;
    });
}

int32_t cg_stats_enable(cg_ctx* ctx, int32_t on) { if (!ctx) return fail(CG_ERR_ARG, "null ctx"); ctx->stats_on = on != 0; return 0; }
int32_t cg_stats(cg_ctx* ctx, cg_stage_times* out, int32_t reset) {
    if (!ctx || !out) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipStreamSynchronize(ctx->sortst));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->aux));
    double* ms[TAG_COUNT] = {&ctx->stats.msm_ms, &ctx->stats.ntt_ms, &ctx->stats.vec_ms, &ctx->stats.spmv_ms,
                             &ctx->stats.msm_sort_ms, &ctx->stats.msm_acc_g1_ms, &ctx->stats.msm_acc_g2_ms, &ctx->stats.msm_reduce_ms};
    uint64_t* calls[TAG_COUNT] = {&ctx->stats.msm_calls, &ctx->stats.ntt_calls, &ctx->stats.vec_calls, &ctx->stats.spmv_calls,
                                  &ctx->stats.msm_sort_calls, &ctx->stats.msm_acc_g1_calls, &ctx->stats.msm_acc_g2_calls, &ctx->stats.msm_reduce_calls};
    for (size_t i = 0; i < ctx->ev_live.size(); i++) {
        EvPair& p = ctx->ev_live[i];
        float t = 0;
        if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) { *ms[p.tag] += t; (*calls[p.tag])++; }
    }
    for (auto& p : ctx->ev_live) ctx->ev_free.push_back(p);
    ctx->ev_live.clear();
    *out = ctx->stats;
    if (reset) ctx->stats = cg_stage_times{};
    return 0;
}

}  // extern "C"
