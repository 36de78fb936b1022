// co-plonk rounds 1-5 (co-circom/co-plonk/src/round{1..5}.rs) over HipDriver, Plonk zkey reader, Keccak transcript
#pragma once
#include "groth16.hpp"

namespace cgh {

// ==================================================================================================== co-plonk, round 1
// First slice of the Plonk prover on the same kernels: [a]_1, [b]_1, [c]_1 = MSM(p_tau, blind(iNTT(wire values))).
// The reference pins the exact result for the blinding b_i = i (co-plonk/src/round1.rs:346-383).
struct PlonkZKey {   // circom-types/src/plonk/zkey.rs:18-42 (the fields round 1 reads)
    Curve curve;
    size_t n_vars = 0, n_public = 0, domain_size = 0, power = 0, n_additions = 0, n_constraints = 0;
    struct Addition { uint32_t id1, id2; Fr f1, f2; };
    std::vector<Addition> additions;
    std::vector<uint32_t> map[3];
    Bytes p_tau;        // domain_size + 6 packed G1 points
    Fr k1, k2;          // verifying key, zkey.rs:328-356
    Bytes vk_g1;        // qm, ql, qr, qo, qc, s1, s2, s3 (8 packed G1 points)
    std::vector<Fr> sigma_eval[3];   // 4 * domain_size evaluations of sigma1..3 (section 12, zkey.rs:116-135,170-180)
    std::vector<Fr> q_eval[5];       // qm, ql, qr, qo, qc on the extended domain (sections 7..11)
    std::vector<std::vector<Fr>> lagrange_eval;   // n_public polynomials on the extended domain (section 13)
    std::vector<Fr> q_coef[5], sigma_coef[3];     // coefficient forms (rounds 4 and 5)
};
static PlonkZKey read_plonk_zkey(int curve_id, const std::string& path) {   // zkey.rs:83-255, header :373-424
    Curve c{curve_id};
    Bytes buf = slurp(path);
    Cursor cur{buf.data(), buf.size()};
    char magic[5] = {0}; cur.bytes(magic, 4);
    if (std::string(magic) != "zkey") throw std::runtime_error("not a zkey file");
    cur.u32();
    uint32_t ns = cur.u32();
    std::map<uint32_t, std::pair<size_t, size_t>> sec;
    for (uint32_t i = 0; i < ns; i++) { uint32_t id = cur.u32(); uint64_t len = cur.u64(); cur.need(len); sec[id] = {cur.off, (size_t)len}; cur.off += len; }
    auto section = [&](uint32_t id) { auto it = sec.find(id); if (it == sec.end()) throw std::runtime_error("missing zkey section"); return Cursor{buf.data() + it->second.first, it->second.second}; };
    if (section(1).u32() != 2) throw std::runtime_error("not a plonk zkey");
    PlonkZKey z; z.curve = c;
    Cursor h = section(2);
    if (h.u32() != c.fq()) throw std::runtime_error("unexpected base field byte size");
    uint64_t q[6] = {0}; h.bytes(q, c.fq());
    if (memcmp(q, MOD_Q[curve_id], c.fq())) throw std::runtime_error("invalid base prime in header");
    if (h.u32() != 32) throw std::runtime_error("unexpected scalar field byte size");
    uint64_t r[4]; h.bytes(r, 32);
    if (memcmp(r, MOD_R[curve_id], 32)) throw std::runtime_error("invalid scalar prime in header");
    z.n_vars = h.u32(); z.n_public = h.u32(); z.domain_size = h.u32(); z.n_additions = h.u32(); z.n_constraints = h.u32();
    if (!z.domain_size || (z.domain_size & (z.domain_size - 1))) throw std::runtime_error("Invalid domain size. Must be power of 2");
    while (((size_t)1 << z.power) < z.domain_size) z.power++;
    h.bytes(z.k1.v, 32); h.bytes(z.k2.v, 32);
    z.vk_g1.resize(8 * c.aff(CG_G1)); h.bytes(z.vk_g1.data(), z.vk_g1.size());
    {
        Cursor sg = section(12);
        for (int k = 0; k < 3; k++) {
            z.sigma_coef[k].resize(z.domain_size); sg.bytes(z.sigma_coef[k].data(), z.domain_size * 32);
            z.sigma_eval[k].resize(4 * z.domain_size);
            sg.bytes(z.sigma_eval[k].data(), 4 * z.domain_size * 32);
        }
    }
    for (int k = 0; k < 5; k++) { Cursor q = section(7 + k); z.q_coef[k].resize(z.domain_size); q.bytes(z.q_coef[k].data(), z.domain_size * 32); z.q_eval[k].resize(4 * z.domain_size); q.bytes(z.q_eval[k].data(), 4 * z.domain_size * 32); }
    { Cursor l = section(13); z.lagrange_eval.resize(z.n_public); for (auto& v : z.lagrange_eval) { l.need(z.domain_size * 32); l.off += z.domain_size * 32; v.resize(4 * z.domain_size); l.bytes(v.data(), 4 * z.domain_size * 32); } }
    { Cursor a = section(3); z.additions.resize(z.n_additions); for (auto& e : z.additions) { e.id1 = a.u32(); e.id2 = a.u32(); a.bytes(e.f1.v, 32); a.bytes(e.f2.v, 32); } }
    for (int k = 0; k < 3; k++) { Cursor m = section(4 + k); z.map[k].resize(z.n_constraints); for (auto& v : z.map[k]) v = m.u32(); }
    { Cursor t = section(14); z.p_tau.resize((z.domain_size + 6) * c.aff(CG_G1)); t.bytes(z.p_tau.data(), z.p_tau.size()); }
    return z;
}

// Keccak-256 (pad 0x01) and the reference's transcript conventions (co-plonk/src/types.rs:122-176): big-endian canonical field
// bytes, 2 * byte_len zero bytes for the point at infinity, challenge = digest as a big-endian integer mod r
class Keccak256 {
    uint64_t a[25]; uint8_t blk[136]; size_t used = 0;
    static uint64_t rotl(uint64_t v, unsigned s) { return s ? (v << s) | (v >> (64 - s)) : v; }
    void f1600() {
        uint64_t lfsr = 1;
        for (int round = 0; round < 24; round++) {
            uint64_t col[5];
            for (int x = 0; x < 5; x++) col[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
            for (int x = 0; x < 5; x++) { const uint64_t d = col[(x + 4) % 5] ^ rotl(col[(x + 1) % 5], 1); for (int y = 0; y < 25; y += 5) a[y + x] ^= d; }
            // rho + pi along the standard lane walk
            int x = 1, y = 0; uint64_t cur = a[1];
            for (int t = 0; t < 24; t++) {
                const int nx = y, ny = (2 * x + 3 * y) % 5;
                const uint64_t nxt = a[nx + 5 * ny];
                a[nx + 5 * ny] = rotl(cur, ((t + 1) * (t + 2) / 2) % 64);
                cur = nxt; x = nx; y = ny;
            }
            for (int yy = 0; yy < 25; yy += 5) { uint64_t r[5]; for (int xx = 0; xx < 5; xx++) r[xx] = a[yy + xx]; for (int xx = 0; xx < 5; xx++) a[yy + xx] = r[xx] ^ (~r[(xx + 1) % 5] & r[(xx + 2) % 5]); }
            for (int j = 0; j < 7; j++) {                                               // iota via the degree-8 LFSR
                const bool bit = lfsr & 1; lfsr = (lfsr << 1) ^ ((lfsr >> 7) * 0x71); lfsr &= 0xff;
                if (bit) a[0] ^= (uint64_t)1 << ((1 << j) - 1);
            }
        }
    }
    void absorb() { for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, blk + 8 * i, 8); a[i] ^= w; } f1600(); used = 0; }
public:
    Keccak256() { memset(a, 0, sizeof a); }
    void update(const uint8_t* p, size_t n) { while (n--) { blk[used++] = *p++; if (used == sizeof blk) absorb(); } }
    void finish(uint8_t out[32]) { memset(blk + used, 0, sizeof blk - used); blk[used] ^= 0x01; blk[sizeof blk - 1] ^= 0x80; absorb(); memcpy(out, a, 32); }
};
class PlonkTranscript {
    const Curve& c; Keccak256 h;
    void be_bytes(const uint64_t* canonical, size_t nbytes) { std::vector<uint8_t> be(nbytes); for (size_t i = 0; i < nbytes; i++) be[nbytes - 1 - i] = (uint8_t)(canonical[i / 8] >> (8 * (i % 8))); h.update(be.data(), nbytes); }
public:
    explicit PlonkTranscript(const Curve& cv) : c(cv) {}
    void add_scalar(const Fr& s) { uint64_t can[4]; CG(cg_fr_to_canonical(c.id, s.v, can, 1)); be_bytes(can, 32); }
    void add_point(const uint8_t* aff) {                                                // packed affine G1, (0,0) = infinity
        if (all_zero_bytes(aff, c.aff(CG_G1))) { std::vector<uint8_t> z(2 * c.fq(), 0); h.update(z.data(), z.size()); return; }
        uint64_t can[12]; CG(cg_fq_to_canonical(c.id, aff, can, 2));
        be_bytes(can, c.fq()); be_bytes(can + c.fq() / 8, c.fq());
    }
    Fr get_challenge() {
        uint8_t d[32]; h.finish(d);
        Fr acc = fr_from_u64(c, 0); const Fr b = fr_from_u64(c, 256);                  // from_be_bytes_mod_order
        for (int i = 0; i < 32; i++) acc = fr_add(c, fr_mul(c, acc, b), fr_from_u64(c, d[i]));
        return acc;
    }
};

// ==================================================================================================== co-plonk, all rounds, any driver
// The five rounds (co-plonk/src/round1..5.rs) written once over share-vector operations: per-component kernels for everything
// linear, the driver's protocols for products of two shared vectors (`mul_vec`), for `array_prod_mul` / `inv_many` (round2.rs:18-41,
// rep3.rs:544-558, shamir.rs:521-535) and for openings.  Plain, REP3 and Shamir run the same code; values that every party reconstructs (commitments,
// evaluations) are functions of the witness and of the opened blinding values only.
class CoPlonk {
public:
    HipDriver& d; const PlonkZKey& z; const cg_bases* tau;
    const Curve c; cg_ctx* ctx; const size_t n, N; const int k;
    Fr zero, one, omega, omega4, w2r;
    FieldShare b[11];
    std::vector<Fr> pub;                       // n_public + 1 values, entry 0 forced to 0 (types.rs:107-109)
    ShareVec buf[3], poly[3], evl[3], poly_z, eval_z, tpart[3];
    Point commit[3], commit_z, commit_t[3], commit_wxi, commit_wxiw;
    Fr beta, gamma, alpha, xi, v[5], ev_a, ev_b, ev_c, ev_s1, ev_s2, ev_zw;
    std::vector<ShareVec> tmp_vecs; std::vector<void*> tmp_ptrs;

    CoPlonk(HipDriver& drv, const PlonkZKey& zk, const cg_bases* p_tau, const std::vector<Fr>& public_inputs, const FieldShare* blind)
        : d(drv), z(zk), tau(p_tau), c(drv.curve), ctx(drv.ctx), n(zk.domain_size), N(4 * zk.domain_size), k(drv.k()), pub(public_inputs) {
        if (pub.size() != z.n_public + 1) throw std::runtime_error("public input length does not match the zkey");
        zero = fr_from_u64(c, 0); one = fr_from_u64(c, 1);
        pub[0] = zero;
        const SnarkjsRoots rt = snarkjs_roots(c);
        omega = rt.roots[z.power]; omega4 = rt.roots[z.power + 2]; w2r = rt.roots[2];
        for (int i = 0; i < 11; i++) b[i] = blind[i];
    }
    ~CoPlonk() {
        release_tmp();
        for (ShareVec* sv : {&buf[0], &buf[1], &buf[2], &poly[0], &poly[1], &poly[2], &evl[0], &evl[1], &evl[2], &poly_z, &eval_z, &tpart[0], &tpart[1], &tpart[2]}) d.free_vec(*sv);
    }
    // ---- share-vector helpers ------------------------------------------------------------------------------------------------
    Fr neg(const Fr& v) const { return fr_sub(c, zero, v); }
    Fr M(const Fr& a, const Fr& x) const { return fr_mul(c, a, x); }
    Fr A(const Fr& a, const Fr& x) const { return fr_add(c, a, x); }
    static uint8_t* at(const ShareVec& s, int j, size_t off = 0) { return (uint8_t*)s.c[j] + off * 32; }
    ShareVec T(size_t len) { ShareVec v = d.alloc_vec(len); tmp_vecs.push_back(v); return v; }          // zeroed temporary, freed by release_tmp
    ShareVec keep(ShareVec v) { tmp_vecs.push_back(v); return v; }
    void* Tp(size_t len) { void* p = d.dalloc(len * 32); tmp_ptrs.push_back(p); return p; }
    void* upload(const std::vector<Fr>& h) { void* p = Tp(h.size()); CG(cg_dev_upload(ctx, p, h.data(), h.size() * 32)); return p; }
    void release_tmp() { for (auto& v : tmp_vecs) d.free_vec(v); tmp_vecs.clear(); for (void* p : tmp_ptrs) CG(cg_dev_free(ctx, p)); tmp_ptrs.clear(); }
    static ShareVec view(const ShareVec& s, size_t off, size_t len) { ShareVec v; v.n = len; for (int j = 0; j < 2; j++) v.c[j] = s.c[j] ? (uint8_t*)s.c[j] + off * 32 : nullptr; return v; }
    void copy(const ShareVec& o, const ShareVec& a, size_t len) { for (int j = 0; j < k; j++) CG(cg_vec_gather_strided_dev(ctx, c.id, o.c[j], a.c[j], len, 0, 1)); }
    void add(const ShareVec& o, const ShareVec& a, const ShareVec& x, size_t len) { for (int j = 0; j < k; j++) CG(cg_vec_add_dev(ctx, c.id, o.c[j], a.c[j], x.c[j], len)); }
    void sub(const ShareVec& o, const ShareVec& a, const ShareVec& x, size_t len) { for (int j = 0; j < k; j++) CG(cg_vec_sub_dev(ctx, c.id, o.c[j], a.c[j], x.c[j], len)); }
    void scale(const ShareVec& o, const ShareVec& a, const Fr& f, size_t len) { for (int j = 0; j < k; j++) CG(cg_vec_affine_dev(ctx, c.id, o.c[j], a.c[j], len, f.v, nullptr)); }   // mul_with_public
    void mulpub(const ShareVec& o, const ShareVec& a, const void* pv, size_t len) { for (int j = 0; j < k; j++) CG(cg_vec_mul_dev(ctx, c.id, o.c[j], a.c[j], pv, len)); }              // by a public vector
    void axpy(const ShareVec& o, const ShareVec& a, const Fr& f, size_t len) { ShareVec t = T(len); scale(t, a, f, len); add(o, o, t, len); }                                           // o += f * a
    void addpub_vec(const ShareVec& o, const ShareVec& a, const void* pv, size_t len) {                                               // add_with_public, element-wise
        const int pc = d.public_component();
        for (int j = 0; j < k; j++) { if (j == pc) CG(cg_vec_add_dev(ctx, c.id, o.c[j], a.c[j], pv, len)); else if (o.c[j] != a.c[j]) CG(cg_vec_gather_strided_dev(ctx, c.id, o.c[j], a.c[j], len, 0, 1)); }
    }
    void addpub_scalar(const ShareVec& o, const ShareVec& a, const Fr& f, size_t len) {
        const int pc = d.public_component();
        for (int j = 0; j < k; j++) { if (j == pc) CG(cg_vec_affine_dev(ctx, c.id, o.c[j], a.c[j], len, one.v, f.v)); else if (o.c[j] != a.c[j]) CG(cg_vec_gather_strided_dev(ctx, c.id, o.c[j], a.c[j], len, 0, 1)); }
    }
    void axpy_pub(const ShareVec& o, const void* pv, const Fr& f, size_t len) {                                                       // o += f * (public vector)
        const int pc = d.public_component(); if (pc < 0) return;
        void* t = Tp(len); CG(cg_vec_affine_dev(ctx, c.id, t, pv, len, f.v, nullptr)); CG(cg_vec_add_dev(ctx, c.id, o.c[pc], o.c[pc], t, len));
    }
    void affine_share(const ShareVec& o, const void* pv, const FieldShare& kk, const FieldShare& dd, size_t len) {                    // o = kk * (public vector) + dd, share-valued kk, dd
        for (int j = 0; j < k; j++) CG(cg_vec_affine_dev(ctx, c.id, o.c[j], pv, len, kk.c[j].v, dd.c[j].v));
    }
    FieldShare get(const ShareVec& s, size_t i) { FieldShare f; f.c[0] = f.c[1] = zero; for (int j = 0; j < k; j++) CG(cg_dev_download(ctx, f.c[j].v, at(s, j, i), 32)); return f; }
    void set(const ShareVec& s, size_t i, const FieldShare& f) { for (int j = 0; j < k; j++) CG(cg_dev_upload(ctx, at(s, j, i), f.c[j].v, 32)); }
    FieldShare fs_sub(const FieldShare& a, const FieldShare& x) const { FieldShare r; for (int j = 0; j < 2; j++) r.c[j] = fr_sub(c, a.c[j], x.c[j]); return r; }
    FieldShare fs_addpub(FieldShare a, const Fr& f) const { const int pc = d.public_component(); if (pc >= 0) a.c[pc] = fr_add(c, a.c[pc], f); return a; }
    ShareVec mul(const ShareVec& a, const ShareVec& x, size_t len) { ShareVec av = view(a, 0, len), xv = view(x, 0, len); return keep(d.mul_vec(av, xv)); }                // mul_vec / mul_many
    Point commit_open(const ShareVec& p, size_t len) {
        if (len > z.domain_size + 6) throw std::runtime_error("polynomial degree too large");
        const PointShare cm = d.msm_public_points(tau, CG_G1, 0, len, p);
        d.verify_received_vectors();       // what a peer sent for this round's products is refused BEFORE anything derived from it is opened (the reference refuses at deserialisation)
        return d.open_point(cm);
    }
    void ntt(const ShareVec& s, size_t len, const Fr& g, bool inverse) { void* ptrs[2] = {s.c[0], s.c[1]}; CG(cg_ntt_dev(ctx, c.id, ptrs, k, len, g.v, inverse ? 1 : 0, nullptr)); }
    // inv_many (rep3.rs:544-558 / plain): element-wise inverse of a shared vector
    ShareVec inv_many(const ShareVec& a, size_t len) {
        ShareVec out = T(len);
        if (d.mode == Mode::Plain) { CG(cg_vec_inverse_dev(ctx, c.id, out.c[0], a.c[0], len)); return out; }
        ShareVec r = keep(d.rand_vec(len));
        void* y = d.mul_open_vec(view(a, 0, len), r); tmp_ptrs.push_back(y);
        CG(cg_vec_inverse_dev(ctx, c.id, y, y, len));                                 // (a zero would make the reference fail with "cannot compute inverse of zero")
        mulpub(out, r, y, len);
        return out;
    }
    // array_prod_mul (round2.rs:18-41): shared prefix products in a constant number of rounds
    ShareVec array_prod_mul(const ShareVec& inp, size_t len) {
        if (d.mode == Mode::Plain) { ShareVec out = T(len); CG(cg_vec_prefix_prod_dev(ctx, c.id, out.c[0], inp.c[0], len)); return out; }
        ShareVec r = keep(d.rand_vec(len + 1));
        ShareVec r_inv = inv_many(r, len + 1);
        ShareVec r_inv0 = T(len);
        const FieldShare first = get(r_inv, 0);
        for (int j = 0; j < k; j++) CG(cg_vec_fill_dev(ctx, c.id, r_inv0.c[j], len, first.c[j].v));
        ShareVec unblind = mul(r_inv0, view(r, 1, len), len);
        ShareVec m = mul(view(r, 0, len), inp, len);
        void* open = d.mul_open_vec(m, view(r_inv, 1, len)); tmp_ptrs.push_back(open);
        CG(cg_vec_prefix_prod_dev(ctx, c.id, open, open, len));
        mulpub(unblind, unblind, open, len);
        return unblind;
    }
    Fr eval_pub_poly(const void* d_poly, size_t len, const Fr& x) {                    // Horner of a public polynomial as a scan
        void* t = Tp(len);
        CG(cg_vec_gather_strided_dev(ctx, c.id, t, d_poly, len, 0, 1));
        CG(cg_vec_distribute_powers_dev(ctx, c.id, t, len, x.v, one.v));
        CG(cg_vec_prefix_sum_dev(ctx, c.id, t, t, len));
        Fr r; CG(cg_dev_download(ctx, r.v, (const uint8_t*)t + (len - 1) * 32, 32));
        return r;
    }
    FieldShare eval_share_poly(const ShareVec& p, size_t len, const Fr& x) {           // evaluate_poly_public (rep3.rs:923-931)
        FieldShare f; f.c[0] = f.c[1] = zero;
        for (int j = 0; j < k; j++) f.c[j] = eval_pub_poly(p.c[j], len, x);
        return f;
    }
    void div_by_zerofier1(const ShareVec& p, size_t len, const Fr& point) {             // round5.rs:97-115 with n = 1; the caller drops the last entry
        const Fr pinv = fr_inv(c, point);
        for (int j = 0; j < k; j++) {
            CG(cg_vec_affine_dev(ctx, c.id, p.c[j], p.c[j], len, neg(pinv).v, nullptr));
            CG(cg_vec_distribute_powers_dev(ctx, c.id, p.c[j], len, point.v, one.v));
            CG(cg_vec_prefix_sum_dev(ctx, c.id, p.c[j], p.c[j], len));
            CG(cg_vec_distribute_powers_dev(ctx, c.id, p.c[j], len, pinv.v, one.v));
        }
    }
    void transcript_point(PlonkTranscript& t, const Point& p) { Bytes a = pt_to_affine(c, p); t.add_point(a.data()); }

    // ---- round 1 (round1.rs:118-312) ---------------------------------------------------------------------------------------------
    FieldShare trivial(const Fr& v) const { FieldShare f; f.c[0] = f.c[1] = zero; const int pc = d.public_component(); if (pc >= 0) f.c[pc] = v; return f; }
    ShareVec extend_witness(const ShareVec& wit) {                                       // calculate_additions (:208-238)
        const size_t n_priv = z.n_vars - z.n_additions - z.n_public - 1;
        std::vector<Fr> ext[2];
        for (int j = 0; j < k; j++) { ext[j].resize(n_priv + z.n_additions); if (n_priv) CG(cg_dev_download(ctx, ext[j].data(), wit.c[j], n_priv * 32)); }
        size_t have = n_priv;
        auto getw = [&](size_t idx) -> FieldShare {
            if (idx <= z.n_public) return trivial(pub[idx]);
            if (idx >= z.n_vars || idx - z.n_public - 1 >= have) throw std::runtime_error("Cannot index into witness " + std::to_string(idx));
            FieldShare f; f.c[0] = f.c[1] = zero; for (int j = 0; j < k; j++) f.c[j] = ext[j][idx - z.n_public - 1];
            return f;
        };
        for (const auto& a : z.additions) { FieldShare w1 = getw(a.id1), w2 = getw(a.id2); for (int j = 0; j < k; j++) ext[j][have] = A(M(a.f1, w1.c[j]), M(a.f2, w2.c[j])); have++; }
        return d.upload_vec(ext[0].data(), k == 2 ? ext[1].data() : nullptr, ext[0].size());
    }
    void round1(const ShareVec& private_witness) {
        const size_t nc = z.n_constraints;
        if (private_witness.n != z.n_vars - z.n_additions - z.n_public - 1) throw std::runtime_error("witness length does not match the zkey");
        ShareVec ext = z.n_additions ? keep(extend_witness(private_witness)) : private_witness;
        void* d_pub = upload(pub);
        std::vector<uint32_t> row_ptr(nc + 1); for (size_t i = 0; i <= nc; i++) row_ptr[i] = (uint32_t)i;
        uint32_t* d_rp = (uint32_t*)Tp((nc + 8) / 8 + 1); CG(cg_dev_upload(ctx, d_rp, row_ptr.data(), (nc + 1) * 4));
        void* d_one = Tp(std::max<size_t>(nc, 1)); CG(cg_vec_fill_dev(ctx, c.id, d_one, std::max<size_t>(nc, 1), one.v));
        uint32_t* d_col = (uint32_t*)Tp((nc + 8) / 8 + 1);
        for (int w = 0; w < 3; w++) {
            if (nc) CG(cg_dev_upload(ctx, d_col, z.map[w].data(), nc * 4));
            poly[w] = d.alloc_vec(n + 2);
            // the wire buffers are gathers: one-entry CSR rows with coefficient one reuse the constraint-evaluation kernel (get_witness' public /
            // private split with the REP3 party asymmetry, lib.rs:113-137)
            CG(cg_spmv_csr_dev(ctx, c.id, d_rp, d_col, d_one, nc, d_pub, (uint32_t)(z.n_public + 1), d.party(), ext.c[0], ext.c[1], poly[w].c[0], poly[w].c[1]));
            buf[w] = d.alloc_vec(n); copy(buf[w], poly[w], n);
            ntt(poly[w], n, omega, true);                                                // :170-172
            evl[w] = d.alloc_vec(N); copy(evl[w], poly[w], n); ntt(evl[w], N, omega4, false);   // :174-177
            const FieldShare &b_hi = b[2 * w], &b_lo = b[2 * w + 1];                     // blind_coefficients (lib.rs:140-158)
            set(poly[w], 0, fs_sub(get(poly[w], 0), b_lo)); set(poly[w], 1, fs_sub(get(poly[w], 1), b_hi));
            set(poly[w], n, b_lo); set(poly[w], n + 1, b_hi);
        }
        PointShare cm[3];
        for (int w = 0; w < 3; w++) cm[w] = d.msm_public_points(tau, CG_G1, 0, n + 2, poly[w]);   // :276-290
        d.verify_received_vectors();
        for (int w = 0; w < 3; w++) commit[w] = d.open_point(cm[w]);                               // open_point_many (:292)
        release_tmp();
    }
    // ---- round 2 (round2.rs:146-298) ---------------------------------------------------------------------------------------------
    void round2() {
        {
            PlonkTranscript t(c);
            for (int i = 0; i < 8; i++) t.add_point(z.vk_g1.data() + i * c.aff(CG_G1));
            for (size_t i = 1; i < pub.size(); i++) t.add_scalar(pub[i]);
            for (int w = 0; w < 3; w++) transcript_point(t, commit[w]);
            beta = t.get_challenge();
            PlonkTranscript t2(c); t2.add_scalar(beta); gamma = t2.get_challenge();
        }
        void* betaw = Tp(n); CG(cg_vec_fill_dev(ctx, c.id, betaw, n, beta.v)); CG(cg_vec_distribute_powers_dev(ctx, c.id, betaw, n, omega.v, one.v));
        void* pv = Tp(n); void* sig = Tp(n);
        const Fr kk[3] = {one, z.k1, z.k2};
        ShareVec num, den;
        for (int w = 0; w < 3; w++) {                                                     // :162-216
            ShareVec f = T(n);
            CG(cg_vec_affine_dev(ctx, c.id, pv, betaw, n, kk[w].v, gamma.v));
            addpub_vec(f, buf[w], pv, n);
            num = w == 0 ? f : mul(num, f, n);
            void* d_sigma = upload(z.sigma_eval[w]);
            CG(cg_vec_gather_strided_dev(ctx, c.id, sig, d_sigma, n, 0, 4));
            CG(cg_vec_affine_dev(ctx, c.id, pv, sig, n, beta.v, gamma.v));
            ShareVec g = T(n);
            addpub_vec(g, buf[w], pv, n);
            den = w == 0 ? g : mul(den, g, n);
        }
        ShareVec num_p = array_prod_mul(num, n), den_p = array_prod_mul(den, n);           // :218-224
        ShareVec den_i = inv_many(den_p, n);                                               // :228
        ShareVec zb = mul(num_p, den_i, n);                                                // :229
        poly_z = d.alloc_vec(n + 3);
        if (n > 1) copy(view(poly_z, 1, n - 1), zb, n - 1);                                // rotate_right(1) (:230)
        copy(poly_z, view(zb, n - 1, 1), 1);
        ntt(poly_z, n, omega, true);                                                       // :235
        eval_z = d.alloc_vec(N); copy(eval_z, poly_z, n); ntt(eval_z, N, omega4, false);   // :238
        set(poly_z, 0, fs_sub(get(poly_z, 0), b[8])); set(poly_z, 1, fs_sub(get(poly_z, 1), b[7])); set(poly_z, 2, fs_sub(get(poly_z, 2), b[6]));
        set(poly_z, n, b[8]); set(poly_z, n + 1, b[7]); set(poly_z, n + 2, b[6]);
        commit_z = commit_open(poly_z, n + 3);                                             // :268-275
        release_tmp();
    }
    // ---- round 3 (round3.rs:234-527) ---------------------------------------------------------------------------------------------
    void round3() {
        if (z.lagrange_eval.empty()) throw std::runtime_error("round 3 needs at least one public input (lagrange[0])");
        { PlonkTranscript t(c); t.add_scalar(beta); t.add_scalar(gamma); transcript_point(t, commit_z); alpha = t.get_challenge(); }   // :498-503
        const Fr alpha2 = M(alpha, alpha), two = fr_from_u64(c, 2);
        const Fr Z1[4] = {zero, A(neg(one), w2r), neg(two), fr_sub(c, neg(one), w2r)};     // get_z1..3 (:203-232)
        const Fr m2w = M(neg(two), w2r);
        const Fr Z2[4] = {zero, m2w, M(two, two), neg(m2w)};
        const Fr tw = M(two, w2r);
        const Fr Z3[4] = {zero, A(two, tw), neg(M(M(two, two), two)), fr_sub(c, two, tw)};
        auto pattern = [&](const Fr* zz) { std::vector<Fr> h(N); for (size_t i = 0; i < N; i++) h[i] = zz[i & 3]; return upload(h); };
        void* z1p = pattern(Z1); void* z2p = pattern(Z2); void* z3p = pattern(Z3);
        const ShareVec &a = evl[0], &bb = evl[1], &cc = evl[2], &ez = eval_z;
        const FieldShare fzero = trivial(zero);
        // the blinding polynomials on the 4n-th roots of unity (:246-256, :307-322)
        void* pw = Tp(N); CG(cg_vec_fill_dev(ctx, c.id, pw, N, one.v)); CG(cg_vec_distribute_powers_dev(ctx, c.id, pw, N, omega4.v, one.v));
        void* pw2 = Tp(N); CG(cg_vec_mul_dev(ctx, c.id, pw2, pw, pw, N));
        void* pww = Tp(N); CG(cg_vec_affine_dev(ctx, c.id, pww, pw, N, omega.v, nullptr));
        void* pww2 = Tp(N); CG(cg_vec_mul_dev(ctx, c.id, pww2, pww, pww, N));
        ShareVec ap = T(N), bp = T(N), cp = T(N), zp = T(N), zwp = T(N), t0 = T(N);
        affine_share(ap, pw, b[0], b[1], N); affine_share(bp, pw, b[2], b[3], N); affine_share(cp, pw, b[4], b[5], N);
        affine_share(zp, pw2, b[6], b[8], N); affine_share(t0, pw, b[7], fzero, N); add(zp, zp, t0, N);
        affine_share(zwp, pww2, b[6], b[8], N); affine_share(t0, pww, b[7], fzero, N); add(zwp, zwp, t0, N);
        ShareVec zw = T(N);                                                                // z(X omega): eval_z rotated by 4 (:324-327)
        copy(zw, view(ez, 4, N - 4), N - 4); copy(view(zw, N - 4, 4), ez, 4);
        // gate constraint (:333-368)
        ShareVec a_b = mul(a, bb, N), a_bp = mul(a, bp, N), ap_b = mul(bb, ap, N), ap_bp = mul(ap, bp, N);
        ShareVec a0 = T(N); add(a0, a_bp, ap_b, N); mulpub(t0, ap_bp, z1p, N); add(a0, a0, t0, N);
        void* q[5]; for (int i = 0; i < 5; i++) q[i] = upload(z.q_eval[i]);
        ShareVec e1 = T(N), e1z = T(N);
        mulpub(e1, a_b, q[0], N); mulpub(t0, a, q[1], N); add(e1, e1, t0, N); mulpub(t0, bb, q[2], N); add(e1, e1, t0, N); mulpub(t0, cc, q[3], N); add(e1, e1, t0, N);
        addpub_vec(e1, e1, q[4], N);
        mulpub(e1z, a0, q[0], N); mulpub(t0, ap, q[1], N); add(e1z, e1z, t0, N); mulpub(t0, bp, q[2], N); add(e1z, e1z, t0, N); mulpub(t0, cp, q[3], N); add(e1z, e1z, t0, N);
        void* l1 = nullptr;
        for (size_t j = 0; j < z.lagrange_eval.size(); j++) {                              // public-input polynomial (:352-358): pi -= L_j * buffer_a[j]
            void* lj = upload(z.lagrange_eval[j]); if (j == 0) l1 = lj;
            const FieldShare aj = get(buf[0], j);
            for (int cpn = 0; cpn < k; cpn++) { CG(cg_vec_affine_dev(ctx, c.id, t0.c[cpn], lj, N, neg(aj.c[cpn]).v, nullptr)); }
            add(e1, e1, t0, N);
        }
        // permutation constraints (:370-418)
        auto mul4 = [&](const ShareVec& Av, const ShareVec& Bv, const ShareVec& Cv, const ShareVec& Dv, const ShareVec& Dp, ShareVec& r, ShareVec& rz) {   // mul4vec + mul4vec_post (:17-72)
            ShareVec AB = mul(Av, Bv, N);
            ShareVec S1 = mul(Av, bp, N); add(S1, S1, mul(ap, Bv, N), N);                  // A B' + A' B
            ShareVec CD = mul(Cv, Dv, N);
            ShareVec S2 = mul(Cv, Dp, N); add(S2, S2, mul(cp, Dv, N), N);                  // C D' + C' D
            ShareVec CpDp = mul(cp, Dp, N);
            r = mul(AB, CD, N);
            rz = mul(S1, CD, N); add(rz, rz, mul(AB, S2, N), N);
            ShareVec r1 = mul(ap_bp, CD, N); add(r1, r1, mul(S1, S2, N), N); add(r1, r1, mul(AB, CpDp, N), N);
            mulpub(t0, r1, z1p, N); add(rz, rz, t0, N);
            ShareVec r2 = mul(S1, CpDp, N); add(r2, r2, mul(ap_bp, S2, N), N);
            mulpub(t0, r2, z2p, N); add(rz, rz, t0, N);
            ShareVec r3 = mul(ap_bp, CpDp, N);
            mulpub(t0, r3, z3p, N); add(rz, rz, t0, N);
        };
        ShareVec e2, e2z, e3, e3z;
        {
            void* pvv = Tp(N);
            ShareVec fa = T(N), fb = T(N), fc = T(N);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, pw, N, beta.v, gamma.v)); addpub_vec(fa, a, pvv, N);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, pw, N, M(beta, z.k1).v, gamma.v)); addpub_vec(fb, bb, pvv, N);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, pw, N, M(beta, z.k2).v, gamma.v)); addpub_vec(fc, cc, pvv, N);
            mul4(fa, fb, fc, ez, zp, e2, e2z);
            ShareVec ga = T(N), gb = T(N), gc = T(N);
            void* sg[3]; for (int i = 0; i < 3; i++) sg[i] = upload(z.sigma_eval[i]);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, sg[0], N, beta.v, gamma.v)); addpub_vec(ga, a, pvv, N);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, sg[1], N, beta.v, gamma.v)); addpub_vec(gb, bb, pvv, N);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, sg[2], N, beta.v, gamma.v)); addpub_vec(gc, cc, pvv, N);
            mul4(ga, gb, gc, zw, zwp, e3, e3z);
        }
        // t = e1 + alpha (e2 - e3) + alpha^2 L1 (z - 1), tz likewise from the blinding parts (:420-441)
        ShareVec Tv = T(N), TZ = T(N);
        sub(t0, e2, e3, N); scale(t0, t0, alpha, N); add(Tv, e1, t0, N);
        addpub_scalar(t0, ez, neg(one), N); mulpub(t0, t0, l1, N); scale(t0, t0, alpha2, N); add(Tv, Tv, t0, N);
        sub(t0, e2z, e3z, N); scale(t0, t0, alpha, N); add(TZ, e1z, t0, N);
        mulpub(t0, zp, l1, N); scale(t0, t0, alpha2, N); add(TZ, TZ, t0, N);
        ntt(Tv, N, omega4, true);                                                          // :442
        scale(view(Tv, 0, n), view(Tv, 0, n), neg(one), n);                                // neg_vec_in_place_limit (:443)
        for (int blk = 1; blk < 4; blk++) sub(view(Tv, blk * n, n), view(Tv, (blk - 1) * n, n), view(Tv, blk * n, n), n);   // division by X^n - 1 (:445-450)
        ntt(TZ, N, omega4, true);
        add(Tv, Tv, TZ, N);                                                                // :453
        const size_t len[3] = {n + 1, n + 1, n + 6};                                        // split (:455-470)
        for (int p = 0; p < 3; p++) { tpart[p] = d.alloc_vec(len[p]); copy(tpart[p], view(Tv, (size_t)p * n, p == 2 ? n + 6 : n), p == 2 ? n + 6 : n); }
        set(tpart[0], n, b[9]);
        set(tpart[1], 0, fs_sub(get(tpart[1], 0), b[9])); set(tpart[1], n, b[10]);
        set(tpart[2], 0, fs_sub(get(tpart[2], 0), b[10]));
        PointShare cm[3];
        for (int p = 0; p < 3; p++) { if (len[p] > z.domain_size + 6) throw std::runtime_error("polynomial degree too large"); cm[p] = d.msm_public_points(tau, CG_G1, 0, len[p], tpart[p]); }
        d.verify_received_vectors();
        for (int p = 0; p < 3; p++) commit_t[p] = d.open_point(cm[p]);                     // :507-522
        release_tmp();
    }
    // ---- rounds 4 and 5 (round4.rs:115-160, round5.rs:143-365) --------------------------------------------------------------------
    void round4() {
        { PlonkTranscript t(c); t.add_scalar(alpha); for (int p = 0; p < 3; p++) transcript_point(t, commit_t[p]); xi = t.get_challenge(); }
        const Fr xiw = M(xi, omega);
        std::vector<FieldShare> sh = {eval_share_poly(poly[0], n + 2, xi), eval_share_poly(poly[1], n + 2, xi), eval_share_poly(poly[2], n + 2, xi), eval_share_poly(poly_z, n + 3, xiw)};
        d.verify_received_vectors();
        const std::vector<Fr> opened = d.open_many(sh);                                    // :131
        ev_a = opened[0]; ev_b = opened[1]; ev_c = opened[2]; ev_zw = opened[3];
        ev_s1 = eval_pub_poly(upload(z.sigma_coef[0]), n, xi); ev_s2 = eval_pub_poly(upload(z.sigma_coef[1]), n, xi);
        release_tmp();
    }
    void round5() {
        { PlonkTranscript t(c); const Fr* items[7] = {&xi, &ev_a, &ev_b, &ev_c, &ev_s1, &ev_s2, &ev_zw}; for (const Fr* e : items) t.add_scalar(*e);
          v[0] = t.get_challenge(); for (int i = 1; i < 5; i++) v[i] = M(v[i - 1], v[0]); }                                          // :338-350
        const size_t len = n + 6;
        Fr xin = xi; for (size_t i = 0; i < z.power; i++) xin = M(xin, xin);               // lib.rs:160-184
        const Fr zh = fr_sub(c, xin, one);
        std::vector<Fr> l; { Fr wv = one; const Fr nn = fr_from_u64(c, (uint64_t)n); for (size_t i = 0; i < std::max<size_t>(1, z.n_public); i++) { l.push_back(M(M(wv, zh), fr_inv(c, M(nn, fr_sub(c, xi, wv))))); wv = M(wv, omega); } }
        Fr eval_pi = zero; for (size_t i = 1; i < pub.size() && i - 1 < l.size(); i++) eval_pi = fr_sub(c, eval_pi, M(l[i - 1], pub[i]));
        const Fr betaxi = M(beta, xi);
        const Fr e2 = M(M(M(A(A(ev_a, betaxi), gamma), A(A(ev_b, M(betaxi, z.k1)), gamma)), A(A(ev_c, M(betaxi, z.k2)), gamma)), alpha);
        const Fr e3 = M(M(M(A(A(ev_a, M(beta, ev_s1)), gamma), A(A(ev_b, M(beta, ev_s2)), gamma)), ev_zw), alpha);
        const Fr e4 = M(M(alpha, alpha), l[0]), e24 = A(e2, e4);
        ShareVec R = T(len);                                                               // compute_r (:143-260)
        axpy(R, poly_z, e24, n + 3);
        void* s_co[3]; for (int i = 0; i < 3; i++) s_co[i] = upload(z.sigma_coef[i]);
        { const Fr f[5] = {M(ev_a, ev_b), ev_a, ev_b, ev_c, one}; for (int i = 0; i < 5; i++) axpy_pub(R, upload(z.q_coef[i]), f[i], n); }
        axpy_pub(R, s_co[2], neg(M(e3, beta)), n);
        axpy(R, tpart[2], neg(M(zh, M(xin, xin))), n + 6); axpy(R, tpart[1], neg(M(zh, xin)), n + 1); axpy(R, tpart[0], neg(zh), n + 1);
        const Fr r0 = fr_sub(c, fr_sub(c, eval_pi, M(e3, A(ev_c, gamma))), e4);
        // compute_wxi (:263-311)
        for (int w = 0; w < 3; w++) axpy(R, poly[w], v[w], n + 2);
        axpy_pub(R, s_co[0], v[3], n); axpy_pub(R, s_co[1], v[4], n);
        const Fr corr = fr_sub(c, r0, A(A(A(A(M(v[0], ev_a), M(v[1], ev_b)), M(v[2], ev_c)), M(v[3], ev_s1)), M(v[4], ev_s2)));
        set(R, 0, fs_addpub(get(R, 0), corr));
        div_by_zerofier1(R, len, xi);
        // compute_wxiw (:314-327)
        ShareVec W = T(n + 3); copy(W, poly_z, n + 3);
        set(W, 0, fs_addpub(get(W, 0), neg(ev_zw)));
        div_by_zerofier1(W, n + 3, M(xi, omega));
        PointShare c1 = d.msm_public_points(tau, CG_G1, 0, len - 1, R), c2 = d.msm_public_points(tau, CG_G1, 0, n + 2, W);   // :351-358
        d.verify_received_vectors();
        commit_wxi = d.open_point(c1); commit_wxiw = d.open_point(c2);
        release_tmp();
    }
};


}  // namespace cgh
