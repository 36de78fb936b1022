// C ABI of the gfx950 co-groth16 backend (include/cogroth16_hip.h) — transform launch logic (twiddle tables, pass plans) and the cg_ntt* entry points
#include "capi_internal.hpp"

namespace {
// ------------------------------------------------------------------------------------------------ NTT
// lo[j] = first * w^j (j < 2^log_lo), hi[j] = w^(j << log_lo) (j < hi_n): w^e * first = lo[e & mask] * hi[e >> log_lo]
template <class Fr>
void host_pow_tables(const Fr& w, const Fr& first, int log_lo, size_t hi_n, std::vector<Fr>& lo, std::vector<Fr>& hi) {
    lo.resize((size_t)1 << log_lo); hi.resize(hi_n);
    Fr acc = first, step = Fr::one();
    for (size_t j = 0; j < lo.size(); j++) { lo[j] = acc; acc = acc * w; step = step * w; }
    Fr h = Fr::one();
    for (size_t j = 0; j < hi_n; j++) { hi[j] = h; h = h * step; }
}

// Twiddle tables depend only on (device, curve, size, generator): contexts of one process share them (three co-located parties, a
// prover serving many proofs).  A table is complete before it is published (the builder synchronises its stream); unused tables
// stay cached up to 1 GiB per process, least recently used first out.
struct SharedTwiddles { void* p; size_t bytes; int refs; uint64_t stamp; };
std::mutex g_tw_mu;
std::map<std::pair<int, TwKey>, SharedTwiddles> g_tw;
uint64_t g_tw_clock = 0;
void* shared_twiddles_acquire(int device, const TwKey& key) {
    std::lock_guard<std::mutex> l(g_tw_mu);
    auto it = g_tw.find({device, key});
    if (it == g_tw.end()) return nullptr;
    it->second.refs++; it->second.stamp = ++g_tw_clock;
    return it->second.p;
}
void* shared_twiddles_publish(int device, const TwKey& key, void* p, size_t bytes) {
    std::lock_guard<std::mutex> l(g_tw_mu);
    auto it = g_tw.find({device, key});
    if (it != g_tw.end()) { hipFree(p); it->second.refs++; it->second.stamp = ++g_tw_clock; return it->second.p; }   // another context was faster
    g_tw[{device, key}] = SharedTwiddles{p, bytes, 1, ++g_tw_clock};
    return p;
}
}  // namespace
void shared_twiddles_release(int device, const TwKey& key) {
    std::lock_guard<std::mutex> l(g_tw_mu);
    auto it = g_tw.find({device, key});
    if (it != g_tw.end() && it->second.refs > 0) it->second.refs--;
    for (;;) {                                            // trim the idle tables
        size_t idle = 0; auto victim = g_tw.end();
        for (auto j = g_tw.begin(); j != g_tw.end(); ++j) if (j->second.refs == 0) { idle += j->second.bytes; if (victim == g_tw.end() || j->second.stamp < victim->second.stamp) victim = j; }
        if (idle <= ((size_t)1 << 30) || victim == g_tw.end()) break;
        hipFree(victim->second.p); g_tw.erase(victim);
    }
}
namespace {

template <class Fr>
int get_twiddles(cg_ctx* ctx, int curve, int log_m, const Fr& w, const Fr** out) {
    TwKey key; key.curve = curve; key.log_m = log_m; memcpy(key.gen, w.v, sizeof key.gen);
    auto it = ctx->twiddles.find(key);
    if (it != ctx->twiddles.end()) { *out = (const Fr*)it->second; return 0; }
    if (void* shared = shared_twiddles_acquire(ctx->device, key)) { ctx->twiddles[key] = shared; *out = (const Fr*)shared; return 0; }
    const size_t m = (size_t)1 << log_m;
    const int log_lo = std::min(11, std::max(0, log_m - 1));
    const size_t hi_n = std::max<size_t>(1, (m / 2) >> log_lo);
    std::vector<Fr> lo, hi;
    host_pow_tables(w, Fr::one(), log_lo, hi_n, lo, hi);
    Fr *d_lo = nullptr, *d_hi = nullptr, *d_tw = nullptr;
    HIPCHK(hip_malloc_flush((void**)&d_lo, lo.size() * sizeof(Fr)));
    HIPCHK(hip_malloc_flush((void**)&d_hi, hi.size() * sizeof(Fr)));
    HIPCHK(hip_malloc_flush((void**)&d_tw, std::max<size_t>(m - 1, 1) * sizeof(Fr)));
    HIPCHK(hipMemcpyAsync(d_lo, lo.data(), lo.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_hi, hi.data(), hi.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    { int rc = launch_build_twiddles<Fr>(ctx->stream, d_tw, m, log_m, d_lo, d_hi, log_lo); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(ctx->stream));   // lo/hi host vectors and temporaries die here
    HIPCHK(hipFree(d_lo)); HIPCHK(hipFree(d_hi));
    d_tw = (Fr*)shared_twiddles_publish(ctx->device, key, d_tw, std::max<size_t>(m - 1, 1) * sizeof(Fr));
    ctx->twiddles[key] = d_tw;
    *out = d_tw;
    return 0;
}

// limb-form table of the lazy passes: tw[i] = 32 * w^bitrev(i), i < m/2 (ntt_kernels.hpp)
template <class Fr>
int get_twiddles_lazy(cg_ctx* ctx, int curve, int log_m, const Fr& w, const void** out) {
    TwKey key; key.curve = curve; key.log_m = log_m; key.kind = 1; memcpy(key.gen, w.v, sizeof key.gen);
    auto it = ctx->twiddles.find(key);
    if (it != ctx->twiddles.end()) { *out = it->second; return 0; }
    if (void* shared = shared_twiddles_acquire(ctx->device, key)) { ctx->twiddles[key] = shared; *out = shared; return 0; }
    const size_t m = (size_t)1 << log_m;
    const int log_lo = std::min(11, std::max(0, log_m - 1));
    const size_t hi_n = std::max<size_t>(1, (m / 2) >> log_lo);
    std::vector<Fr> lo, hi;
    host_pow_tables(w, Fr::one(), log_lo, hi_n, lo, hi);
    Fr c32 = Fr::one(); for (int i = 0; i < 5; i++) c32 = c32 + c32;
    Fr *d_lo = nullptr, *d_hi = nullptr; void* d_tw = nullptr;
    const size_t bytes = lazy29_bytes(std::max<size_t>(m / 2, 1));
    HIPCHK(hip_malloc_flush((void**)&d_lo, lo.size() * sizeof(Fr)));
    HIPCHK(hip_malloc_flush((void**)&d_hi, hi.size() * sizeof(Fr)));
    HIPCHK(hip_malloc_flush(&d_tw, bytes));
    HIPCHK(hipMemcpyAsync(d_lo, lo.data(), lo.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_hi, hi.data(), hi.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    { int rc = launch_build_twiddles_lazy<Fr>(ctx->stream, d_tw, m, log_m, d_lo, d_hi, log_lo, c32); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(d_lo)); HIPCHK(hipFree(d_hi));
    d_tw = shared_twiddles_publish(ctx->device, key, d_tw, bytes);
    ctx->twiddles[key] = d_tw;
    *out = d_tw;
    return 0;
}

// natural-order limb-form table of the decimation-in-time passes: tw[e] = 32 * w^e, e < m/2 (ntt_kernels.hpp, k_ntt_dit_pass)
template <class Fr>
int get_twiddles_lazy_natural(cg_ctx* ctx, int curve, int log_m, const Fr& w, const void** out) {
    TwKey key; key.curve = curve; key.log_m = log_m; key.kind = 2; memcpy(key.gen, w.v, sizeof key.gen);
    auto it = ctx->twiddles.find(key);
    if (it != ctx->twiddles.end()) { *out = it->second; return 0; }
    if (void* shared = shared_twiddles_acquire(ctx->device, key)) { ctx->twiddles[key] = shared; *out = shared; return 0; }
    const size_t m = (size_t)1 << log_m;
    const int log_lo = std::min(11, std::max(0, log_m - 1));
    const size_t hi_n = std::max<size_t>(1, (m / 2) >> log_lo);
    std::vector<Fr> lo, hi;
    host_pow_tables(w, Fr::one(), log_lo, hi_n, lo, hi);
    Fr c32 = Fr::one(); for (int i = 0; i < 5; i++) c32 = c32 + c32;
    Fr *d_lo = nullptr, *d_hi = nullptr; void* d_tw = nullptr;
    const size_t bytes = lazy29_bytes(std::max<size_t>(m / 2, 1));
    HIPCHK(hip_malloc_flush((void**)&d_lo, lo.size() * sizeof(Fr)));
    HIPCHK(hip_malloc_flush((void**)&d_hi, hi.size() * sizeof(Fr)));
    HIPCHK(hip_malloc_flush(&d_tw, bytes));
    HIPCHK(hipMemcpyAsync(d_lo, lo.data(), lo.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_hi, hi.data(), hi.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    { int rc = launch_build_twiddles_lazy_natural<Fr>(ctx->stream, d_tw, m, d_lo, d_hi, log_lo, c32); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(d_lo)); HIPCHK(hipFree(d_hi));
    d_tw = shared_twiddles_publish(ctx->device, key, d_tw, bytes);
    ctx->twiddles[key] = d_tw;
    *out = d_tw;
    return 0;
}

// tables with lo[j] = scale * g^j, hi[j] = g^(j << log_lo), covering exponents < 2^log_m
template <class Fr>
int get_coset_tables(cg_ctx* ctx, int curve, int log_m, const Fr& g, const Fr& scale, CosetTables* out) {
    CosetKey key; key.k.curve = curve; key.k.log_m = log_m; memcpy(key.k.gen, g.v, sizeof key.k.gen); memcpy(key.scale, scale.v, sizeof key.scale);
    auto it = ctx->cosets.find(key);
    if (it != ctx->cosets.end()) { *out = it->second; return 0; }
    const size_t m = (size_t)1 << log_m;
    const int log_lo = std::min(11, log_m);
    const size_t hi_n = std::max<size_t>(1, m >> log_lo);
    std::vector<Fr> lo, hi;
    host_pow_tables(g, scale, log_lo, hi_n, lo, hi);
    CosetTables t; t.log_lo = log_lo;
    HIPCHK(hip_malloc_flush(&t.lo, lo.size() * sizeof(Fr)));
    HIPCHK(hip_malloc_flush(&t.hi, hi.size() * sizeof(Fr)));
    HIPCHK(hipMemcpy(t.lo, lo.data(), lo.size() * sizeof(Fr), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(t.hi, hi.data(), hi.size() * sizeof(Fr), hipMemcpyHostToDevice));
    if (ctx->cosets.size() >= 64) {   // callers that scale by per-proof challenges would otherwise grow the cache without bound
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (auto& kv : ctx->cosets) { hipFree(kv.second.lo); hipFree(kv.second.hi); }
        ctx->cosets.clear();
    }
    ctx->cosets[key] = t;
    *out = t;
    return 0;
}

struct NttPass { int s0, k, t; };
std::vector<NttPass> ntt_plan(int log_m, int tile_log = NTT_TILE_LOG) {
    std::vector<NttPass> plan;
    const int k_last = std::min(log_m, tile_log);
    const int rest = log_m - k_last;
    int s0 = 0;
    if (rest > 0) {
        const int np = (rest + 6) / 7;
        for (int i = 0; i < np; i++) {
            int k = rest / np + (i < rest % np ? 1 : 0);
            plan.push_back({s0, k, tile_log - k});       // lo_bits >= tile_log here, so t = tile_log - k fits
            s0 += k;
        }
    }
    plan.push_back({s0, k_last, 0});
    return plan;
}

template <class Fr>
int ntt_run(cg_ctx* ctx, int curve, void* const* d_vecs, int k, size_t n, const Fr& gen, bool inverse, const Fr* coset, size_t arena_off) {
    const int log_m = log2_floor(n);
    if (((size_t)1 << log_m) != n) return fail(CG_ERR_ARG, "NTT length must be a power of two");
    if (k < 1 || k > NTT_MAX_VECS) return fail(CG_ERR_ARG, "k out of range");
    if (n == 1) return 0;
    const Fr w = inverse ? fp_inverse(gen) : gen;
    static const bool legacy = tune_env("CG_NTT_DIF") != nullptr;                 // A/B knob: the canonical DIF passes
    if (!legacy) {
        // lazy Cooley-Tukey passes (ntt_kernels.hpp): packed vectors -> limb-form scratch -> ... -> permutation back into the vectors,
        // which multiplies by 32 * (1/m) * coset power (32: the lazy core divides by 2^261, the ABI's R is 2^256)
        if (!inverse && coset) return fail(CG_ERR_ARG, "coset_gen is only supported with inverse != 0");
        const void* twl = nullptr;
        int rc = get_twiddles_lazy<Fr>(ctx, curve, log_m, w, &twl);
        if (rc) return rc;
        NttVecs data{}, tmp{};
        for (int j = 0; j < k; j++) { data.p[j] = d_vecs[j]; tmp.p[j] = ctx->ntt_arena.base + arena_off + (size_t)j * lazy29_bytes(n); }
        hipStream_t st = ctx->stream;
        bool first = true;
        static const int lazy_tile = [] { const char* e = tune_env("CG_NTT_TILE"); const int v = e ? atoi(e) : NTT_TILE_LOG_LAZY; return std::min(NTT_TILE_LOG, std::max(8, v)); }();   // tuning knob
        for (const NttPass& p : ntt_plan(log_m, lazy_tile)) { rc = launch_ntt_ct_pass<Fr>(st, first, first ? data : tmp, tmp, k, n, log_m, p.s0, p.k, p.t, twl); if (rc) return rc; first = false; }
        Fr scale32 = Fr::one(); for (int i = 0; i < 5; i++) scale32 = scale32 + scale32;
        if (inverse) {
            uint32_t e[Fr::N] = {0}; e[log_m / 32] = 1u << (log_m % 32);
            Fr nn; for (int i = 0; i < Fr::N; i++) nn.v[i] = e[i];
            scale32 = scale32 * fp_inverse(nn.to_mont());
        }
        CosetTables t;
        const Fr* d_scale = nullptr; const Fr* c_lo = nullptr; const Fr* c_hi = nullptr; int log_lo = 0;
        if (coset) { rc = get_coset_tables<Fr>(ctx, curve, log_m, *coset, scale32, &t); if (rc) return rc; c_lo = (const Fr*)t.lo; c_hi = (const Fr*)t.hi; log_lo = t.log_lo; }
        else { rc = get_coset_tables<Fr>(ctx, curve, 0, Fr::one(), scale32, &t); if (rc) return rc; d_scale = (const Fr*)t.lo; }
        return launch_bitrev_finish_lazy<Fr>(st, data, tmp, k, n, log_m, d_scale, c_lo, c_hi, log_lo);
    }
    const Fr* tw = nullptr;
    int rc = get_twiddles<Fr>(ctx, curve, log_m, w, &tw);
    if (rc) return rc;
    NttVecs data{}, tmp{};
    for (int j = 0; j < k; j++) { data.p[j] = d_vecs[j]; tmp.p[j] = ctx->ntt_arena.base + arena_off + (size_t)j * n * sizeof(Fr); }
    hipStream_t st = ctx->stream;
    {   // first pass reads the caller's vectors and writes the scratch copies; later passes run in the scratch copies
        bool first = true;
        for (const NttPass& p : ntt_plan(log_m)) { rc = launch_ntt_dif_pass<Fr>(st, first ? data : tmp, tmp, k, n, log_m, p.s0, p.k, p.t, tw); if (rc) return rc; first = false; }
    }
    const Fr* d_scale = nullptr; const Fr* c_lo = nullptr; const Fr* c_hi = nullptr; int log_lo = 0;
    if (inverse) {
        Fr ninv = Fr::one();   // n^-1: halve log_m times  (x/2 = (x + (x odd ? p : 0)) >> 1 in Montgomery form as well)
        {
            uint32_t e[Fr::N] = {0}; e[log_m / 32] = 1u << (log_m % 32);
            Fr nn; for (int i = 0; i < Fr::N; i++) nn.v[i] = e[i];
            ninv = fp_inverse(nn.to_mont());
        }
        CosetTables t;
        if (coset) { rc = get_coset_tables<Fr>(ctx, curve, log_m, *coset, ninv, &t); if (rc) return rc; c_lo = (const Fr*)t.lo; c_hi = (const Fr*)t.hi; log_lo = t.log_lo; }
        else { rc = get_coset_tables<Fr>(ctx, curve, 0, Fr::one(), ninv, &t); if (rc) return rc; d_scale = (const Fr*)t.lo; }
    } else if (coset) return fail(CG_ERR_ARG, "coset_gen is only supported with inverse != 0");
    // the permutation brings the result back: tmp -> data (natural order), fused with 1/m and the coset powers
    return launch_bitrev_scale<Fr>(st, data, tmp, k, n, log_m, d_scale, c_lo, c_hi, log_lo);
}

// v <- NTT_w( g^i * (iNTT_w v)_i ): the inverse transform's passes leave the coefficients bit-reversed in limb-form scratch, the
// decimation-in-time passes take them from there (scaling by (1/m) g^i on the way in) and write the natural-order result
template <class Fr>
int ntt_coset_pair_run(cg_ctx* ctx, int curve, void* const* d_vecs, int k, size_t n, const Fr& gen, const Fr& coset, size_t arena_off) {
    const int log_m = log2_floor(n);
    if (((size_t)1 << log_m) != n) return fail(CG_ERR_ARG, "NTT length must be a power of two");
    if (k < 1 || k > NTT_MAX_VECS) return fail(CG_ERR_ARG, "k out of range");
    if (n == 1) return 0;                                                       // both transforms and g^0 are the identity
    const void* tw_inv = nullptr; const void* tw_fwd = nullptr;
    int rc = get_twiddles_lazy<Fr>(ctx, curve, log_m, fp_inverse(gen), &tw_inv); if (rc) return rc;
    rc = get_twiddles_lazy_natural<Fr>(ctx, curve, log_m, gen, &tw_fwd); if (rc) return rc;
    Fr c32 = Fr::one(); for (int i = 0; i < 5; i++) c32 = c32 + c32;
    uint32_t e[Fr::N] = {0}; e[log_m / 32] = 1u << (log_m % 32);
    Fr nn; for (int i = 0; i < Fr::N; i++) nn.v[i] = e[i];
    CosetTables ct;
    rc = get_coset_tables<Fr>(ctx, curve, log_m, coset, c32 * fp_inverse(nn.to_mont()), &ct); if (rc) return rc;
    NttVecs data{}, tmp{};
    for (int j = 0; j < k; j++) { data.p[j] = d_vecs[j]; tmp.p[j] = ctx->ntt_arena.base + arena_off + (size_t)j * lazy29_bytes(n); }
    hipStream_t st = ctx->stream;
    static const int lazy_tile = [] { const char* e_ = tune_env("CG_NTT_TILE"); const int v = e_ ? atoi(e_) : NTT_TILE_LOG_LAZY; return std::min(NTT_TILE_LOG, std::max(8, v)); }();
    const std::vector<NttPass> plan = ntt_plan(log_m, lazy_tile);
    bool first = true;
    for (const NttPass& p : plan) { rc = launch_ntt_ct_pass<Fr>(st, first, first ? data : tmp, tmp, k, n, log_m, p.s0, p.k, p.t, tw_inv); if (rc) return rc; first = false; }
    for (size_t i = plan.size(); i-- > 0;) {
        const NttPass& p = plan[i];
        rc = launch_ntt_dit_pass<Fr>(st, i + 1 == plan.size(), i == 0, data, tmp, k, n, log_m, p.s0, p.k, p.t, tw_fwd, (const Fr*)ct.lo, (const Fr*)ct.hi, ct.log_lo, c32);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

extern "C" {
// ---------------------------------------------------------------------------------------------------- NTT
int32_t cg_ntt_dev(cg_ctx* ctx, int32_t curve, void* const* d_vecs, int32_t k, size_t n, const void* h_group_gen, int32_t inverse, const void* h_coset_gen) {
    if (!ctx || !d_vecs || !h_group_gen) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        Fr gen, cos; copy_in(gen, h_group_gen);
        if (h_coset_gen) copy_in(cos, h_coset_gen);
        if (n > 1) { int rc = ensure_ntt_arena(ctx, (size_t)k * lazy29_bytes(n)); if (rc) return rc; }
        StatScope ss(ctx, TAG_NTT);
        return ntt_run<Fr>(ctx, curve, d_vecs, k, n, gen, inverse != 0, h_coset_gen ? &cos : nullptr, 0);
    });
}
int32_t cg_ntt_coset_pair_dev(cg_ctx* ctx, int32_t curve, void* const* d_vecs, int32_t k, size_t n, const void* h_group_gen, const void* h_coset_gen) {
    if (!ctx || !d_vecs || !h_group_gen || !h_coset_gen) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    static const bool two_calls = tune_env("CG_NTT_NO_PAIR") != nullptr;         // A/B knob: the two separate transforms
    if (two_calls) { int rc = cg_ntt_dev(ctx, curve, d_vecs, k, n, h_group_gen, 1, h_coset_gen); return rc ? rc : cg_ntt_dev(ctx, curve, d_vecs, k, n, h_group_gen, 0, nullptr); }
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        Fr gen, cos; copy_in(gen, h_group_gen); copy_in(cos, h_coset_gen);
        if (n > 1) { int rc = ensure_ntt_arena(ctx, (size_t)k * lazy29_bytes(n)); if (rc) return rc; }
        StatScope ss(ctx, TAG_NTT);
        return ntt_coset_pair_run<Fr>(ctx, curve, d_vecs, k, n, gen, cos, 0);
    });
}
int32_t cg_ntt(cg_ctx* ctx, int32_t curve, void* const* h_vecs, int32_t k, size_t n, const void* h_group_gen, int32_t inverse, const void* h_coset_gen) {
    if (!ctx || !h_vecs) return fail(CG_ERR_ARG, "null argument");
    if (k < 1 || k > NTT_MAX_VECS) return fail(CG_ERR_ARG, "k out of range");
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<void*> d(k, nullptr);
    for (int j = 0; j < k; j++) { HIPCHK(hip_malloc_flush(&d[j], std::max<size_t>(n * 32, 16))); HIPCHK(hipMemcpyAsync(d[j], h_vecs[j], n * 32, hipMemcpyHostToDevice, ctx->stream)); }
    int rc = cg_ntt_dev(ctx, curve, d.data(), k, n, h_group_gen, inverse, h_coset_gen);
    if (!rc) for (int j = 0; j < k; j++) { hipError_t e = hipMemcpyAsync(h_vecs[j], d[j], n * 32, hipMemcpyDeviceToHost, ctx->stream); if (e != hipSuccess) rc = fail(CG_ERR_HIP, hipGetErrorString(e)); }
    hipStreamSynchronize(ctx->stream);
    for (int j = 0; j < k; j++) hipFree(d[j]);
    return rc;
}

// distribute_powers_and_mul_by_const on its own (rep3.rs:681-688): the coset tables are the transforms'
int32_t cg_vec_distribute_powers_dev(cg_ctx* ctx, int32_t curve, void* d_v, size_t n, const void* h_g, const void* h_c) {
    if (!ctx || !d_v || !h_g || !h_c) return fail(CG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return 0;
    return with_fr(curve, [&](auto tag) -> int {
        typedef decltype(tag) Fr;
        Fr g, c; copy_in(g, h_g); copy_in(c, h_c);
        int log_m = log2_floor(n); if (((size_t)1 << log_m) < n) log_m++;
        CosetTables t;
        int rc = get_coset_tables<Fr>(ctx, curve, log_m, g, c, &t);
        if (rc) return rc;
        StatScope ss(ctx, TAG_VEC);
        return launch_distribute_powers<Fr>(ctx->stream, (Fr*)d_v, n, (const Fr*)t.lo, (const Fr*)t.hi, t.log_lo);
    });
}
}  // extern "C"
