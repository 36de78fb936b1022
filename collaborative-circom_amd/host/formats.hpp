// zkey / wtns readers on a mapped file and the snarkjs evaluation domain (circom-types/src/groth16/zkey.rs:139-316, binfile.rs:52-97, witness.rs:51-91; co-circom-snarks/src/lib.rs:208-221, groth16.rs:57-77)
#pragma once
#include "base.hpp"

namespace cgh {

// ---- file formats ----------------------------------------------------------------------------------------------------
struct Cursor {
    const uint8_t* p; size_t n, off = 0;
    void need(size_t k) const { if (off > n || k > n - off) throw std::runtime_error("unexpected end of section"); }   // no wrap-around for a 64-bit length read from the file
    uint32_t u32() { need(4); uint32_t x; memcpy(&x, p + off, 4); off += 4; return x; }
    uint64_t u64() { need(8); uint64_t x; memcpy(&x, p + off, 8); off += 8; return x; }
    void bytes(void* d, size_t k) { need(k); memcpy(d, p + off, k); off += k; }
};
static Bytes slurp(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    Bytes b((size_t)n);
    if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); throw std::runtime_error("short read"); }
    fclose(f);
    return b;
}
// zkey -> device fast path: the file is mapped, the point sections are handed to cg_bases_register where they lie (one
// host->device copy, no intermediate buffers); only the small header points and the coefficient section are decoded on the host.
struct MappedFile {
    const uint8_t* p = nullptr; size_t n = 0;
    explicit MappedFile(const std::string& path) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat st; if (fstat(fd, &st) != 0) { close(fd); throw std::runtime_error("cannot stat " + path); }
        n = (size_t)st.st_size;
        if (n) { void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); if (m == MAP_FAILED) { close(fd); throw std::runtime_error("cannot map " + path); } p = (const uint8_t*)m; }
        close(fd);
    }
    ~MappedFile() { if (p) munmap((void*)p, n); }
    MappedFile(const MappedFile&) = delete; MappedFile& operator=(const MappedFile&) = delete;
};
struct View {   // a section of the mapped file, with the read-only part of the std::vector interface the prover uses
    const uint8_t* p = nullptr; size_t n = 0;
    const uint8_t* data() const { return p; } size_t size() const { return n; }
    const uint8_t* begin() const { return p; } const uint8_t* end() const { return p + n; }
};

static const uint64_t MOD_R[2][4] = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
                                     {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}};
static const uint64_t MOD_Q[2][6] = {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull, 0, 0},
                                     {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull}};

struct ZKey {   // zkey.rs:48-71; points kept in the packed on-disk form (x||y Montgomery, (0,0) = infinity) the ABI accepts directly
    Curve curve;
    size_t n_vars = 0, n_public = 0, domain_size = 0, pow = 0, num_constraints = 0;
    Bytes alpha_g1, beta_g1, delta_g1, beta_g2, gamma_g2, delta_g2;
    std::shared_ptr<MappedFile> file;                     // keeps the views below alive
    View ic, a_query, b_g1_query, b_g2_query, l_query, h_query;
    std::vector<uint32_t> row_ptr[2], col[2];
    std::vector<Fr> coeff[2];
};

static ZKey read_zkey(int curve_id, const std::string& path, bool header_only = false) {   // header_only: sizes and counts, no matrix values
    Curve c{curve_id};
    auto mf = std::make_shared<MappedFile>(path);
    struct { const uint8_t* p; size_t n; const uint8_t* data() const { return p; } size_t size() const { return n; } } buf{mf->p, mf->n};
    Cursor cur{buf.data(), buf.size()};
    char magic[5] = {0}; cur.bytes(magic, 4);
    if (std::string(magic) != "zkey") throw std::runtime_error("not a zkey file");
    cur.u32();
    uint32_t ns = cur.u32();
    std::map<uint32_t, std::pair<size_t, size_t>> sec;
    for (uint32_t i = 0; i < ns; i++) { uint32_t id = cur.u32(); uint64_t len = cur.u64(); cur.need(len); sec[id] = {cur.off, (size_t)len}; cur.off += len; }
    auto section = [&](uint32_t id) { auto it = sec.find(id); if (it == sec.end()) throw std::runtime_error("missing zkey section"); return Cursor{buf.data() + it->second.first, it->second.second}; };
    ZKey z; z.curve = c; z.file = mf;
    {   // header, zkey.rs:258-316
        Cursor h = section(2);
        if (h.u32() != c.fq()) throw std::runtime_error("unexpected base field byte size");
        uint64_t q[6] = {0}; h.bytes(q, c.fq());
        if (memcmp(q, MOD_Q[curve_id], c.fq())) throw std::runtime_error("invalid base prime in header");
        if (h.u32() != 32) throw std::runtime_error("unexpected scalar field byte size");
        uint64_t r[4]; h.bytes(r, 32);
        if (memcmp(r, MOD_R[curve_id], 32)) throw std::runtime_error("invalid scalar prime in header");
        z.n_vars = h.u32(); z.n_public = h.u32(); z.domain_size = h.u32();
        if (!z.domain_size || (z.domain_size & (z.domain_size - 1))) throw std::runtime_error("domain size must be a power of two");
        if (z.n_vars <= z.n_public) throw std::runtime_error("invalid data: n_vars must exceed n_public");
        while (((size_t)1 << z.pow) < z.domain_size) z.pow++;
        auto g = [&](int grp) { Bytes b(c.aff(grp)); h.bytes(b.data(), b.size()); return b; };
        z.alpha_g1 = g(CG_G1); z.beta_g1 = g(CG_G1); z.beta_g2 = g(CG_G2); z.gamma_g2 = g(CG_G2); z.delta_g1 = g(CG_G1); z.delta_g2 = g(CG_G2);
    }
    auto pts = [&](uint32_t id, size_t n, int grp) { Cursor s = section(id); s.need(n * c.aff(grp)); return View{s.p, n * c.aff(grp)}; };
    z.ic = pts(3, z.n_public + 1, CG_G1); z.a_query = pts(5, z.n_vars, CG_G1); z.b_g1_query = pts(6, z.n_vars, CG_G1);
    z.b_g2_query = pts(7, z.n_vars, CG_G2); z.l_query = pts(8, z.n_vars - z.n_public - 1, CG_G1); z.h_query = pts(9, z.domain_size, CG_G1);
    {   // section 4, zkey.rs:184-204: records (u32 matrix, u32 row, u32 signal, 32 B value); value on disk = v*R^2, one Montgomery
        // reduction gives the Montgomery form of v (traits.rs:65-67).  Three passes over the mapped records: last row, row counts,
        // CSR fill with the raw values; then the values are reduced in place, in parallel slices.
        Cursor s = section(4);
        const uint32_t ncoef = s.u32();
        s.need((size_t)ncoef * 44);
        const uint8_t* rec = s.p + s.off;
        auto word = [&](size_t i, int k) { uint32_t v; memcpy(&v, rec + i * 44 + 4 * k, 4); return v; };
        uint32_t max_row = 0;
        for (size_t i = 0; i < ncoef; i++) { if (word(i, 0) > 1) throw std::runtime_error("bad matrix id"); max_row = std::max(max_row, word(i, 1)); }
        if (ncoef == 0 || max_row < z.n_public) throw std::runtime_error("invalid data: coefficient section has no rows beyond the public inputs");
        z.num_constraints = (size_t)max_row - z.n_public;
        for (int m = 0; m < 2; m++) z.row_ptr[m].assign(z.num_constraints + 1, 0);
        for (size_t i = 0; i < ncoef; i++) { const uint32_t row = word(i, 1); if (row < z.num_constraints) z.row_ptr[word(i, 0)][row + 1]++; }
        std::vector<uint32_t> fill[2];
        for (int m = 0; m < 2; m++) {
            for (size_t i = 0; i < z.num_constraints; i++) z.row_ptr[m][i + 1] += z.row_ptr[m][i];
            z.col[m].resize(z.row_ptr[m].back()); z.coeff[m].resize(z.col[m].size());
            fill[m].assign(z.row_ptr[m].begin(), z.row_ptr[m].end() - 1);
        }
        if (!header_only) {
            for (size_t i = 0; i < ncoef; i++) {
                const uint32_t m = word(i, 0), row = word(i, 1);
                if (row >= z.num_constraints) continue;
                const uint32_t k = fill[m][row]++;
                if (word(i, 2) >= z.n_vars) throw std::runtime_error("invalid data: matrix column index beyond n_vars");   // the device mat-vec indexes the witness with it
                z.col[m][k] = word(i, 2); memcpy(z.coeff[m][k].v, rec + i * 44 + 12, 32);
            }
            const int nthreads = (int)std::min<size_t>(8, std::max<size_t>(1, ncoef / 65536));
            std::vector<std::thread> th; std::vector<int> rc(2 * nthreads, 0);
            for (int m = 0; m < 2; m++) for (int t = 0; t < nthreads; t++) th.emplace_back([&, m, t] {
                const size_t n = z.coeff[m].size(), lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
                if (hi > lo) rc[m * nthreads + t] = cg_fr_to_canonical(c.id, z.coeff[m].data() + lo, z.coeff[m].data() + lo, hi - lo);
            });
            for (auto& x : th) x.join();
            for (int r : rc) if (r) die("cg_fr_to_canonical");
        }
    }
    return z;
}

static std::vector<Fr> read_wtns(int curve_id, const std::string& path) {   // witness.rs:51-91
    Bytes buf = slurp(path);
    Cursor c{buf.data(), buf.size()};
    char magic[5] = {0}; c.bytes(magic, 4);
    if (std::string(magic) != "wtns") throw std::runtime_error("not a wtns file");
    if (c.u32() > 2) throw std::runtime_error("wtns version not supported");
    if (c.u32() > 2) throw std::runtime_error("invalid section number");
    c.u32(); c.u64();
    if (c.u32() != 32) throw std::runtime_error("wrong scalar field");
    uint64_t mod[4]; c.bytes(mod, 32);
    if (memcmp(mod, MOD_R[curve_id], 32)) throw std::runtime_error("wrong scalar field");
    uint32_t n = c.u32();
    c.u32(); c.u64();
    std::vector<Fr> raw(n), out(n);
    c.bytes(raw.data(), (size_t)n * 32);
    CG(cg_fr_from_canonical(curve_id, raw.data(), out.data(), n));
    return out;
}

// co-circom-snarks/src/lib.rs:208-221 + groth16.rs:57-77
struct Domain { size_t m; int log_m; Fr omega, coset_g; };
// roots[i] = primitive 2^i-th root of unity derived from the smallest quadratic non-residue (co-circom-snarks/src/lib.rs:208-221)
struct SnarkjsRoots { Fr q; std::vector<Fr> roots; int two_adicity; };
static SnarkjsRoots snarkjs_roots(const Curve& c) {
    const uint64_t* r = MOD_R[c.id];
    uint64_t t[4] = {r[0] - 1, r[1], r[2], r[3]};
    int s = 0;
    while (!(t[0] & 1)) { for (int i = 0; i < 4; i++) t[i] = (t[i] >> 1) | (i < 3 ? t[i + 1] << 63 : 0); s++; }
    uint64_t half[4] = {r[0] - 1, r[1], r[2], r[3]};
    for (int i = 0; i < 4; i++) half[i] = (half[i] >> 1) | (i < 3 ? half[i + 1] << 63 : 0);
    const Fr one = fr_from_u64(c, 1), minus_one = fr_sub(c, fr_from_u64(c, 0), one);
    Fr q = one;
    while (!fr_eq(fr_pow(c, q, half, 4), minus_one)) q = fr_add(c, q, one);       // smallest quadratic non-residue
    std::vector<Fr> roots(s + 1);
    roots[0] = fr_pow(c, q, t, 4);
    for (int i = 1; i <= s; i++) roots[i] = fr_mul(c, roots[i - 1], roots[i - 1]);
    return SnarkjsRoots{q, std::vector<Fr>(roots.rbegin(), roots.rend()), s};
}
// (computed once per curve: the search for the non-residue is half a dozen 254-bit exponentiations, ~80 us through the ABI's field calls — and a
// party's proof asks for its domain three times)
static const SnarkjsRoots& snarkjs_roots_cached(const Curve& c) {
    static std::mutex mu; static std::map<int, SnarkjsRoots> table;
    std::lock_guard<std::mutex> l(mu);
    auto it = table.find(c.id);
    if (it == table.end()) it = table.emplace(c.id, snarkjs_roots(c)).first;
    return it->second;                                                               // (entries are never removed: the reference stays valid)
}
static Domain groth16_domain(const Curve& c, size_t pow, size_t num_constraints, size_t num_inputs) {
    const SnarkjsRoots& rt = snarkjs_roots_cached(c);
    Domain d; d.m = 1; d.log_m = 0;
    while (d.m < num_constraints + num_inputs) { d.m <<= 1; d.log_m++; }
    d.omega = rt.roots[pow];
    d.coset_g = rt.two_adicity == d.log_m ? fr_mul(c, rt.q, rt.q) : rt.roots[d.log_m + 1];
    return d;
}

}  // namespace cgh
