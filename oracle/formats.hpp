// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// Readers for the circom/snarkjs binary formats feeding the Groth16 path, restating
//   `/root/reference/co-circom/circom-types/src/binfile.rs:52-97`      (container: magic, version, sections)
//   `/root/reference/co-circom/circom-types/src/groth16/zkey.rs:139-316` (sections 2..9, matrices truncation :200-204)
//   `/root/reference/co-circom/circom-types/src/traits.rs:57-67,107-155` (Montgomery point decode, value*R^2 coeff decode)
//   `/root/reference/co-circom/circom-types/src/witness.rs:51-91`        (wtns: canonical LE values)
#pragma once
#include "curves.hpp"
#include "poly.hpp"
#include <cstdio>
#include <map>

namespace orc {

struct BinSections {
    std::string magic;
    uint32_t version = 0;
    std::map<uint32_t, std::vector<uint8_t>> sec;
};

static inline std::vector<uint8_t> read_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)n);
    if (n && fread(buf.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); throw std::runtime_error("short read " + path); }
    fclose(f);
    return buf;
}

struct Cursor {
    const uint8_t* p; size_t n, off = 0;
    Cursor(const std::vector<uint8_t>& v) : p(v.data()), n(v.size()) {}
    void need(size_t k) const { if (off + k > n) throw std::runtime_error("unexpected end of section"); }
    uint32_t u32() { need(4); uint32_t x; memcpy(&x, p + off, 4); off += 4; return x; }
    uint64_t u64() { need(8); uint64_t x; memcpy(&x, p + off, 8); off += 8; return x; }
    void bytes(void* dst, size_t k) { need(k); memcpy(dst, p + off, k); off += k; }
};

static inline BinSections read_binfile(const std::string& path) {
    auto buf = read_file(path);
    Cursor c(buf);
    BinSections out;
    char magic[5] = {0}; c.bytes(magic, 4); out.magic = magic;
    out.version = c.u32();
    uint32_t ns = c.u32();
    for (uint32_t i = 0; i < ns; i++) {
        uint32_t id = c.u32(); uint64_t len = c.u64();
        c.need(len);
        if (out.sec.count(id)) throw std::runtime_error("duplicate section");
        out.sec[id] = std::vector<uint8_t>(c.p + c.off, c.p + c.off + len);
        c.off += len;
    }
    return out;
}

template <class F>
static F read_mont(Cursor& c) {   // already Montgomery on disk (traits.rs:57-63)
    uint64_t l[F::N]; c.bytes(l, sizeof l);
    if (raw_cmp<F::N>(l, F::K.p) >= 0) throw std::runtime_error("field element not reduced");
    return F::from_mont_limbs(l);
}
template <class Fq>
static AffineT<Fq> read_g1(Cursor& c) {
    Fq x = read_mont<Fq>(c), y = read_mont<Fq>(c);
    if (x.is_zero() && y.is_zero()) return AffineT<Fq>::infinity();
    return {x, y, false};
}
template <class Fq>
static AffineT<Fp2T<Fq>> read_g2(Cursor& c) {
    Fq x0 = read_mont<Fq>(c), x1 = read_mont<Fq>(c), y0 = read_mont<Fq>(c), y1 = read_mont<Fq>(c);
    Fp2T<Fq> x = {x0, x1}, y = {y0, y1};
    if (x.is_zero() && y.is_zero()) return AffineT<Fp2T<Fq>>::infinity();
    return {x, y, false};
}

template <class C>
struct ZKey {
    typedef typename C::Fr Fr;
    typedef typename C::G1::Affine G1A;
    typedef typename C::G2::Affine G2A;
    size_t n_vars = 0, n_public = 0, domain_size = 0, pow = 0, num_constraints = 0;
    G1A alpha_g1, beta_g1, delta_g1;
    G2A beta_g2, gamma_g2, delta_g2;
    std::vector<G1A> ic, a_query, b_g1_query, l_query, h_query;
    std::vector<G2A> b_g2_query;
    // CSR matrices A (index 0) and B (index 1), rows = num_constraints, column = signal index
    std::vector<uint32_t> row_ptr[2], col[2];
    std::vector<Fr> coeff[2];
};

template <class C>
static ZKey<C> read_zkey(const std::string& path, bool check_points = true) {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq;
    C::init();
    BinSections bf = read_binfile(path);
    if (bf.magic != "zkey") throw std::runtime_error("not a zkey");
    ZKey<C> z;
    {   // header (section 2)
        Cursor c(bf.sec.at(2));
        uint32_t n8q = c.u32();
        if (n8q != Fq::N * 8) throw std::runtime_error("unexpected base field byte size");
        uint64_t q[Fq::N]; c.bytes(q, sizeof q);
        if (raw_cmp<Fq::N>(q, Fq::K.p) != 0) throw std::runtime_error("invalid base prime in header");
        uint32_t n8r = c.u32();
        if (n8r != Fr::N * 8) throw std::runtime_error("unexpected scalar field byte size");
        uint64_t r[Fr::N]; c.bytes(r, sizeof r);
        if (raw_cmp<Fr::N>(r, Fr::K.p) != 0) throw std::runtime_error("invalid scalar prime in header");
        z.n_vars = c.u32(); z.n_public = c.u32(); z.domain_size = c.u32();
        if (z.domain_size == 0 || (z.domain_size & (z.domain_size - 1))) throw std::runtime_error("domain size not a power of two");
        z.pow = (size_t)log2_exact(z.domain_size);
        z.alpha_g1 = read_g1<Fq>(c); z.beta_g1 = read_g1<Fq>(c);
        z.beta_g2 = read_g2<Fq>(c); z.gamma_g2 = read_g2<Fq>(c);
        z.delta_g1 = read_g1<Fq>(c); z.delta_g2 = read_g2<Fq>(c);
    }
    auto g1vec = [&](uint32_t id, size_t n) { Cursor c(bf.sec.at(id)); std::vector<typename C::G1::Affine> v(n); for (auto& p : v) p = read_g1<Fq>(c); return v; };
    z.ic = g1vec(3, z.n_public + 1);
    z.a_query = g1vec(5, z.n_vars);
    z.b_g1_query = g1vec(6, z.n_vars);
    { Cursor c(bf.sec.at(7)); z.b_g2_query.resize(z.n_vars); for (auto& p : z.b_g2_query) p = read_g2<Fq>(c); }
    z.l_query = g1vec(8, z.n_vars - z.n_public - 1);
    z.h_query = g1vec(9, z.domain_size);
    if (check_points) {
        auto chk1 = [&](const std::vector<typename C::G1::Affine>& v) { for (auto& p : v) if (!C::G1::on_curve(p)) throw std::runtime_error("G1 point not on curve"); };
        chk1(z.ic); chk1(z.a_query); chk1(z.b_g1_query); chk1(z.l_query); chk1(z.h_query);
        for (auto& p : z.b_g2_query) if (!C::G2::on_curve(p)) throw std::runtime_error("G2 point not on curve");
    }
    {   // section 4: coefficients; value on disk = v*R^2 -> Montgomery rep of v is disk*R^-1 (traits.rs:65-67)
        Cursor c(bf.sec.at(4));
        uint32_t ncoef = c.u32();
        struct E { uint32_t m, row, sig; Fr v; };
        std::vector<E> es(ncoef);
        uint32_t max_row = 0;
        Fr raw_one; memset(raw_one.v, 0, sizeof raw_one.v); raw_one.v[0] = 1;
        for (auto& e : es) {
            e.m = c.u32(); e.row = c.u32(); e.sig = c.u32();
            Fr disk = read_mont<Fr>(c);
            e.v = disk * raw_one;
            if (e.m > 1) throw std::runtime_error("bad matrix id");
            max_row = std::max(max_row, e.row);
        }
        z.num_constraints = (size_t)max_row - z.n_public;
        for (int m = 0; m < 2; m++) {
            z.row_ptr[m].assign(z.num_constraints + 1, 0);
            for (auto& e : es) if (e.m == (uint32_t)m && e.row < z.num_constraints) z.row_ptr[m][e.row + 1]++;
            for (size_t i = 0; i < z.num_constraints; i++) z.row_ptr[m][i + 1] += z.row_ptr[m][i];
            z.col[m].resize(z.row_ptr[m][z.num_constraints]);
            z.coeff[m].resize(z.col[m].size());
            std::vector<uint32_t> fill(z.row_ptr[m].begin(), z.row_ptr[m].end() - 1);
            for (auto& e : es) if (e.m == (uint32_t)m && e.row < z.num_constraints) {   // file order kept within a row
                uint32_t k = fill[e.row]++;
                z.col[m][k] = e.sig; z.coeff[m][k] = e.v;
            }
        }
    }
    return z;
}

template <class Fr>
static std::vector<Fr> read_wtns(const std::string& path) {
    auto buf = read_file(path);
    Cursor c(buf);
    char magic[5] = {0}; c.bytes(magic, 4);
    if (std::string(magic) != "wtns") throw std::runtime_error("not a wtns file");
    uint32_t version = c.u32();
    if (version > 2) throw std::runtime_error("wtns version not supported");
    uint32_t nsec = c.u32();
    if (nsec > 2) throw std::runtime_error("invalid section number");
    c.u32(); c.u64();
    uint32_t n8 = c.u32();
    if (n8 != Fr::N * 8) throw std::runtime_error("wrong scalar field");
    uint64_t mod[Fr::N]; c.bytes(mod, sizeof mod);
    if (raw_cmp<Fr::N>(mod, Fr::K.p) != 0) throw std::runtime_error("wrong scalar field");
    uint32_t n = c.u32();
    c.u32(); c.u64();
    std::vector<Fr> out(n);
    for (auto& v : out) { uint64_t l[Fr::N]; c.bytes(l, sizeof l); v = Fr::from_canonical(l); }
    return out;
}

}  // namespace orc
