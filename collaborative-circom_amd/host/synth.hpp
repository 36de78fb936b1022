// synthetic satisfiable circuit + valid Groth16 CRS written as .zkey / .wtns (bench and test tooling)
#pragma once
#include "groth16.hpp"

namespace cgh {

// ---- synthetic satisfiable circuit + valid Groth16 CRS (bench / test tooling; SURVEY.md §8d "synthetic R1CS generator") --------------
// Writes a snarkjs-format .zkey (sections 1-9, the layout read_zkey above parses: circom-types/src/groth16/zkey.rs:139-316) and a
// .wtns (witness.rs:51-91), so that sessions and file -> proof runs have a real file of any size to work on: the shipped fixtures stop
// at 213 constraints.  n_public = 1, num_constraints = m - 2, n_vars = m = domain size; constraint j:
//     (ca_j * w[j+1]) * (cb_j * w[sb_j]) = w[j+2],   sb_j = 1 + (7 j + 3) mod (j + 1)  (<= j + 1: the witness is computed forward).
// CRS from seeded toxic waste (tau, alpha, beta, gamma, delta): polynomial evaluations on the host (field arithmetic through the
// ABI's cg_fr_op, on a few threads), the five point tables by fixed-base batch multiplication on the GPU (cg_bases_from_scalars).
// Conventions the prover relies on (groth16.rs:141-204): section 4 carries the rows A[nc + i] = w_i for i <= n_public, and
//     h_query[i] = [ (tau^2m - 1) g w^i / (2 m delta (tau - g w^i)) ]_1,   g = w_2m:
// H = (AB - C)/Z is interpolated on the odd coset gH, where Z = g^m - 1 = -2, so the prover's h_i = (AB - C)(g w^i) needs no division.
static void batch_inverse(const Curve& c, std::vector<Fr>& v) {           // Montgomery's trick per slice; no zero elements
    parallel_for(v.size(), [&](size_t lo, size_t hi) {
        if (hi <= lo) return;
        std::vector<Fr> pre(hi - lo);
        Fr acc = fr_from_u64(c, 1);
        for (size_t i = lo; i < hi; i++) { pre[i - lo] = acc; acc = fr_mul(c, acc, v[i]); }
        Fr inv = fr_inv(c, acc);
        for (size_t i = hi; i-- > lo;) { const Fr t = fr_mul(c, inv, pre[i - lo]); inv = fr_mul(c, inv, v[i]); v[i] = t; }
    });
}
struct SplitMix { uint64_t s; uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); } };
static Fr random_nonzero_fr(const Curve& c, SplitMix& g) {
    for (;;) {
        Fr raw; for (int i = 0; i < 4; i++) raw.v[i] = g.next();
        raw.v[3] &= c.id == CG_BN254 ? 0x3fffffffffffffffull : 0x7fffffffffffffffull;
        bool lt = false, gt = false;
        for (int i = 3; i >= 0 && !lt && !gt; i--) { if (raw.v[i] < MOD_R[c.id][i]) lt = true; else if (raw.v[i] > MOD_R[c.id][i]) gt = true; }
        if (!lt || !(raw.v[0] | raw.v[1] | raw.v[2] | raw.v[3])) continue;
        Fr m; CG(cg_fr_from_canonical(c.id, raw.v, m.v, 1));
        return m;
    }
}
struct SectionWriter {     // sections are streamed: a 2^22-constraint zkey is 2 GB
    FILE* f;
    SectionWriter(const std::string& path, const char* magic, uint32_t version, uint32_t nsec) {
        f = fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + path);
        put(magic, 4); u32(version); u32(nsec);
    }
    ~SectionWriter() { if (f) fclose(f); }
    void put(const void* p, size_t n) { if (n && fwrite(p, 1, n, f) != n) throw std::runtime_error("short write"); }
    void u32(uint32_t x) { put(&x, 4); }
    void u64(uint64_t x) { put(&x, 8); }
    void begin(uint32_t id, uint64_t bytes) { u32(id); u64(bytes); }
    void close() { if (f && fclose(f) != 0) { f = nullptr; throw std::runtime_error("close failed"); } f = nullptr; }
};
static void synth_circuit(int device, int curve_id, int log_m, uint64_t seed, const std::string& zkey_path, const std::string& wtns_path) {
    if (log_m < 2 || log_m > 26) throw std::runtime_error("log_m out of range");
    const Curve c{curve_id};
    const size_t m = (size_t)1 << log_m, nc = m - 2, n_pub = 1, n_vars = m, n_inp = n_pub + 1;
    SplitMix rng{seed * 0x2545f4914f6cdd1dull + 0x1234567};
    // circuit + witness (a serial chain by construction)
    std::vector<Fr> ca(nc), cb(nc), w(n_vars);
    std::vector<uint32_t> sb(nc);
    for (size_t j = 0; j < nc; j++) { ca[j] = random_nonzero_fr(c, rng); cb[j] = random_nonzero_fr(c, rng); sb[j] = (uint32_t)(1 + (7 * j + 3) % (j + 1)); }
    w[0] = fr_from_u64(c, 1); w[1] = random_nonzero_fr(c, rng);
    for (size_t j = 0; j < nc; j++) w[j + 2] = fr_mul(c, fr_mul(c, ca[j], w[j + 1]), fr_mul(c, cb[j], w[sb[j]]));
    // toxic waste, Lagrange values of the domain at tau
    const Fr tau = random_nonzero_fr(c, rng), alpha = random_nonzero_fr(c, rng), beta = random_nonzero_fr(c, rng), gamma = random_nonzero_fr(c, rng), delta = random_nonzero_fr(c, rng);
    const Domain dom = groth16_domain(c, (size_t)log_m, nc, n_inp);
    const Fr one = fr_from_u64(c, 1);
    Fr tau_m = tau; for (int i = 0; i < log_m; i++) tau_m = fr_mul(c, tau_m, tau_m);
    std::vector<Fr> wpow(m), lag(m), hexp(m);
    {   // w^j by slices: each slice starts from w^lo (square-and-multiply) and runs a product chain
        parallel_for(m, [&](size_t lo, size_t hi) {
            if (hi <= lo) return;
            uint64_t e[1] = {lo};
            Fr acc = fr_pow(c, dom.omega, e, 1);
            for (size_t j = lo; j < hi; j++) { wpow[j] = acc; acc = fr_mul(c, acc, dom.omega); }
        });
    }
    parallel_for(m, [&](size_t lo, size_t hi) { for (size_t j = lo; j < hi; j++) { lag[j] = fr_sub(c, tau, wpow[j]); hexp[j] = fr_sub(c, tau, fr_mul(c, dom.coset_g, wpow[j])); } });
    batch_inverse(c, lag); batch_inverse(c, hexp);
    const Fr zt_over_m = fr_mul(c, fr_sub(c, tau_m, one), fr_inv(c, fr_from_u64(c, (uint64_t)m)));
    const Fr hfac = fr_mul(c, fr_mul(c, fr_sub(c, fr_mul(c, tau_m, tau_m), one), fr_inv(c, fr_mul(c, fr_from_u64(c, 2 * (uint64_t)m), delta))), dom.coset_g);
    parallel_for(m, [&](size_t lo, size_t hi) { for (size_t j = lo; j < hi; j++) { lag[j] = fr_mul(c, fr_mul(c, zt_over_m, wpow[j]), lag[j]); hexp[j] = fr_mul(c, fr_mul(c, hfac, wpow[j]), hexp[j]); } });
    { std::vector<Fr>().swap(wpow); }
    // u_i = sum_j A[j][i] L_j(tau), v_i, and the C column (C[j][j+2] = 1)
    const Fr zero = fr_sub(c, one, one);
    std::vector<Fr> u(n_vars, zero), v(n_vars, zero), lic(n_vars);
    parallel_for(nc, [&](size_t lo, size_t hi) { for (size_t j = lo; j < hi; j++) u[j + 1] = fr_mul(c, ca[j], lag[j]); });
    for (size_t j = 0; j < nc; j++) v[sb[j]] = fr_add(c, v[sb[j]], fr_mul(c, cb[j], lag[j]));     // colliding targets: serial
    for (size_t i = 0; i < n_inp; i++) u[i] = fr_add(c, u[i], lag[nc + i]);
    const Fr ginv = fr_inv(c, gamma), dinv = fr_inv(c, delta);
    parallel_for(n_vars, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            Fr t = fr_add(c, fr_mul(c, beta, u[i]), fr_mul(c, alpha, v[i]));
            if (i >= 2) t = fr_add(c, t, lag[i - 2]);                                         // C column: w_i = L_(i-2) for i >= 2
            lic[i] = fr_mul(c, t, i <= n_pub ? ginv : dinv);
        }
    });
    // group elements on the GPU
    CtxGuard cg; if (cg_ctx_create(device, &cg.ctx)) die("cg_ctx_create");
    cg_ctx* ctx = cg.ctx;
    auto table = [&](const std::vector<Fr>& sc, int group) {
        DevBufGuard d{ctx, nullptr};
        CG(cg_dev_alloc(ctx, sc.size() * 32, &d.p));
        CG(cg_dev_upload(ctx, d.p, sc.data(), sc.size() * 32));
        cg_bases* b = nullptr; CG(cg_bases_from_scalars(ctx, c.id, group, d.p, sc.size(), &b));
        Bytes out(sc.size() * c.aff(group));
        const int rc = cg_bases_download(ctx, b, 0, sc.size(), out.data());
        cg_bases_release(b);
        if (rc) die("cg_bases_download");
        return out;
    };
    auto g1 = [&](const Fr& k) { return pt_to_affine(c, pt_mul(c, pt_generator(c, CG_G1), k)); };
    auto g2 = [&](const Fr& k) { return pt_to_affine(c, pt_mul(c, pt_generator(c, CG_G2), k)); };
    const uint32_t ncoef = (uint32_t)(2 * nc + n_inp);
    SectionWriter zk(zkey_path, "zkey", 1, 9);
    zk.begin(1, 4); zk.u32(1);                                                               // protocol: groth16
    zk.begin(2, 4 + c.fq() + 4 + 32 + 12 + 3 * c.aff(CG_G1) + 3 * c.aff(CG_G2));
    zk.u32((uint32_t)c.fq()); zk.put(MOD_Q[c.id], c.fq()); zk.u32(32); zk.put(MOD_R[c.id], 32);
    zk.u32((uint32_t)n_vars); zk.u32((uint32_t)n_pub); zk.u32((uint32_t)m);
    { Bytes a1 = g1(alpha), b1 = g1(beta), b2 = g2(beta), c2 = g2(gamma), d1 = g1(delta), d2 = g2(delta);
      zk.put(a1.data(), a1.size()); zk.put(b1.data(), b1.size()); zk.put(b2.data(), b2.size()); zk.put(c2.data(), c2.size()); zk.put(d1.data(), d1.size()); zk.put(d2.data(), d2.size()); }
    Bytes l_all = table(lic, CG_G1);
    { std::vector<Fr>().swap(lic); }
    zk.begin(3, n_inp * c.aff(CG_G1)); zk.put(l_all.data(), n_inp * c.aff(CG_G1));
    zk.begin(4, 4 + (uint64_t)ncoef * 44); zk.u32(ncoef);
    {   // value on disk = v * R^2: the Montgomery form of the Montgomery form (traits.rs:57-67 reduces once)
        auto rec = [&](uint32_t mat, uint32_t row, uint32_t sig, const Fr& val) { Fr d; CG(cg_fr_from_canonical(c.id, val.v, d.v, 1)); zk.u32(mat); zk.u32(row); zk.u32(sig); zk.put(d.v, 32); };
        for (size_t j = 0; j < nc; j++) { rec(0, (uint32_t)j, (uint32_t)(j + 1), ca[j]); rec(1, (uint32_t)j, sb[j], cb[j]); }
        for (size_t i = 0; i < n_inp; i++) rec(0, (uint32_t)(nc + i), (uint32_t)i, one);
    }
    { Bytes t = table(u, CG_G1); zk.begin(5, t.size()); zk.put(t.data(), t.size()); }
    { Bytes t = table(v, CG_G1); zk.begin(6, t.size()); zk.put(t.data(), t.size()); }
    { Bytes t = table(v, CG_G2); zk.begin(7, t.size()); zk.put(t.data(), t.size()); }
    zk.begin(8, (n_vars - n_inp) * c.aff(CG_G1)); zk.put(l_all.data() + n_inp * c.aff(CG_G1), (n_vars - n_inp) * c.aff(CG_G1));
    { Bytes t = table(hexp, CG_G1); zk.begin(9, t.size()); zk.put(t.data(), t.size()); }
    zk.close();
    SectionWriter wt(wtns_path, "wtns", 2, 2);
    wt.begin(1, 4 + 32 + 4); wt.u32(32); wt.put(MOD_R[c.id], 32); wt.u32((uint32_t)n_vars);
    wt.begin(2, (uint64_t)n_vars * 32);
    { std::vector<Fr> can(n_vars); CG(cg_fr_to_canonical(c.id, w.data(), can.data(), n_vars)); wt.put(can.data(), n_vars * 32); }
    wt.close();
}

}  // namespace cgh
