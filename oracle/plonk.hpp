// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// co-plonk, round 1 (the wire-polynomial commitments): the first slice of SURVEY.md §8 f-2.  It exercises the SAME two hot
// operations as Groth16 (iNTT with the snarkjs root, variable-base MSM over `p_tau`) and — unlike Groth16 — the reference pins
// its exact output: `co-plonk/src/round1.rs:346-383` hard-codes [a]_1, [b]_1, [c]_1 for test_vectors/Plonk/bn254/multiplier2 with
// the deterministic blinding b_i = i.  Restates:
//   `/root/reference/co-circom/circom-types/src/plonk/zkey.rs:83-255,328-425`  (container sections 2..6 and 14, header)
//   `/root/reference/co-circom/co-plonk/src/types.rs:59-100,101-116`          (domain root = snarkjs roots[pow]; public_inputs[0] := 0)
//   `/root/reference/co-circom/co-plonk/src/lib.rs:113-158`                   (get_witness, blind_coefficients)
//   `/root/reference/co-circom/co-plonk/src/round1.rs:118-206,208-238,260-312` (wire polynomials, additions, commitments)
#pragma once
#include "formats.hpp"
#include "ec.hpp"

namespace orc {

template <class C>
struct PlonkZKey {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq;
    size_t n_vars = 0, n_public = 0, domain_size = 0, power = 0, n_additions = 0, n_constraints = 0;
    struct Addition { uint32_t id1, id2; Fr f1, f2; };
    std::vector<Addition> additions;
    std::vector<uint32_t> map_a, map_b, map_c;
    std::vector<AffineT<Fq>> p_tau;          // domain_size + 6 points (zkey.rs:149-151)
};

template <class C>
static PlonkZKey<C> read_plonk_zkey(const std::string& path) {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq;
    BinSections bf = read_binfile(path);
    if (bf.magic != "zkey") throw std::runtime_error("not a zkey file");
    auto sec = [&](uint32_t id) -> const std::vector<uint8_t>& { auto it = bf.sec.find(id); if (it == bf.sec.end()) throw std::runtime_error("missing zkey section"); return it->second; };
    { Cursor c(sec(1)); if (c.u32() != 2) throw std::runtime_error("not a plonk zkey (protocol id != 2)"); }
    PlonkZKey<C> z;
    {   // header, zkey.rs:373-424
        Cursor h(sec(2));
        if (h.u32() != Fq::N * 8) throw std::runtime_error("unexpected base field byte size");
        uint64_t q[Fq::N]; h.bytes(q, sizeof q);
        if (raw_cmp<Fq::N>(q, Fq::K.p) != 0) throw std::runtime_error("invalid base prime in header");
        if (h.u32() != Fr::N * 8) throw std::runtime_error("unexpected scalar field byte size");
        uint64_t r[Fr::N]; h.bytes(r, sizeof r);
        if (raw_cmp<Fr::N>(r, Fr::K.p) != 0) throw std::runtime_error("invalid scalar prime in header");
        z.n_vars = h.u32(); z.n_public = h.u32(); z.domain_size = h.u32(); z.n_additions = h.u32(); z.n_constraints = h.u32();
        if (!z.domain_size || (z.domain_size & (z.domain_size - 1))) throw std::runtime_error("invalid domain size");
        while (((size_t)1 << z.power) < z.domain_size) z.power++;
    }
    { Cursor c(sec(3)); z.additions.resize(z.n_additions); for (auto& a : z.additions) { a.id1 = c.u32(); a.id2 = c.u32(); a.f1 = read_mont<Fr>(c); a.f2 = read_mont<Fr>(c); } }
    auto id_map = [&](uint32_t id) { Cursor c(sec(id)); std::vector<uint32_t> m(z.n_constraints); for (auto& v : m) v = c.u32(); return m; };
    z.map_a = id_map(4); z.map_b = id_map(5); z.map_c = id_map(6);
    { Cursor c(sec(14)); z.p_tau.resize(z.domain_size + 6); for (auto& p : z.p_tau) p = read_g1<Fq>(c); }
    return z;
}

// plain driver; `full_witness` is the Groth16-style witness (leading constant one); blind[0..6) = b_1..b_6 of round 1
template <class C>
static std::vector<AffineT<typename C::Fq>> plonk_round1_plain(const PlonkZKey<C>& z, const std::vector<typename C::Fr>& full_witness, const typename C::Fr* blind,
                                                                 std::vector<typename C::Fr>* polys_out = nullptr) {
    typedef typename C::Fr Fr; typedef typename C::G1 G1;
    if (full_witness.size() + z.n_additions != z.n_vars) throw std::runtime_error("witness length does not match the zkey");
    std::vector<Fr> w(full_witness);
    w[0] = Fr::zero();                                                        // types.rs:107-109: snarkjs writes 0 for the constant
    auto get = [&](size_t idx) -> Fr { if (idx >= z.n_vars || idx >= w.size()) throw std::runtime_error("corrupted witness index"); return w[idx]; };   // lib.rs:113-137
    for (const auto& a : z.additions) w.push_back(get(a.id1) * a.f1 + get(a.id2) * a.f2);                                 // round1.rs:208-238
    const size_t n = z.domain_size;
    const Fr omega = roots_of_unity<Fr>().roots[z.power];                      // types.rs:70-84
    std::vector<AffineT<typename C::Fq>> commits;
    const std::vector<uint32_t>* maps[3] = {&z.map_a, &z.map_b, &z.map_c};
    for (int k = 0; k < 3; k++) {
        std::vector<Fr> buf(n, Fr::zero());
        for (size_t i = 0; i < z.n_constraints; i++) buf[i] = get((*maps[k])[i]);                                         // round1.rs:134-157
        ntt_inverse(buf.data(), n, omega);                                                                                // :170-172
        std::vector<Fr> poly(buf);
        const Fr b_hi = blind[2 * k], b_lo = blind[2 * k + 1];                 // coeff_rev = [b_hi, b_lo]; reversed: b_lo first (lib.rs:140-158)
        poly[0] = poly[0] - b_lo; poly[1] = poly[1] - b_hi;
        poly.push_back(b_lo); poly.push_back(b_hi);
        if (poly.size() > z.p_tau.size()) throw std::runtime_error("polynomial degree too large");
        commits.push_back(msm_naive<G1>(z.p_tau.data(), poly.data(), poly.size()).to_affine());                          // round1.rs:276-290
        if (polys_out) polys_out->insert(polys_out->end(), poly.begin(), poly.end());
    }
    return commits;
}

}  // namespace orc
