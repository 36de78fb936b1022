// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// Short-Weierstrass curve arithmetic y^2 = x^3 + b (a = 0: BN254 and BLS12-381, G1 and G2) restating what the
// reference obtains from ark-ec 0.4.2 (`/root/reference/Cargo.toml:35`, not vendored):
//   * `short_weierstrass::Projective{x,y,z}` is Jacobian, z == 0 <=> infinity
//     (result type of `MSMProvider::msm_public_points`, `/root/reference/mpc-core/src/traits.rs:561-568`)
//   * `C::msm_unchecked(points, scalars)` call sites `/root/reference/mpc-core/src/protocols/rep3.rs:942-943`,
//     `shamir.rs:1035`, `plain.rs:414`, `co-circom/co-groth16/src/groth16.rs:220`
// The MSM below restates the published ark-ec 0.4.2 `msm_bigint` (window size c = 3 if n < 32 else
// floor(log2(n)*69/100)+2, unsigned c-bit digits, 2^c-1 Jacobian buckets per window, running-sum bucket
// reduction, Horner fold with c doublings, unit scalars short-cut in window 0, windows processed in parallel).
// Which algorithm is used does not affect the value: the affine form of the sum is unique.
#pragma once
#include "ff.hpp"
#include <thread>
#include <atomic>

namespace orc {

template <class F>
struct AffineT {
    F x, y;
    bool inf;
    static AffineT infinity() { return {F::zero(), F::zero(), true}; }
    bool operator==(const AffineT& o) const {
        if (inf || o.inf) return inf == o.inf;
        return x == o.x && y == o.y;
    }
};

template <class F, int TAG>
struct JacT {
    typedef F Field;
    typedef AffineT<F> Affine;
    F x, y, z;
    static inline F B;   // curve coefficient b (Montgomery)

    static JacT infinity() { return {F::one(), F::one(), F::zero()}; }
    static JacT from_affine(const Affine& a) { return a.inf ? infinity() : JacT{a.x, a.y, F::one()}; }
    bool is_inf() const { return z.is_zero(); }
    JacT neg() const { return {x, -y, z}; }

    static bool on_curve(const Affine& a) {
        if (a.inf) return true;
        return a.y.sqr() == a.x.sqr() * a.x + B;
    }

    // dbl-2009-l (a = 0)
    JacT dbl() const {
        if (is_inf()) return *this;
        F A = x.sqr(), Bq = y.sqr(), C = Bq.sqr();
        F D = ((x + Bq).sqr() - A - C).dbl();
        F E = A.dbl() + A;
        F Fq = E.sqr();
        F X3 = Fq - D.dbl();
        F Y3 = E * (D - X3) - C.dbl().dbl().dbl();
        F Z3 = (y * z).dbl();
        return {X3, Y3, Z3};
    }
    // add-2007-bl with the exceptional cases handled
    JacT add(const JacT& o) const {
        if (is_inf()) return o;
        if (o.is_inf()) return *this;
        F Z1Z1 = z.sqr(), Z2Z2 = o.z.sqr();
        F U1 = x * Z2Z2, U2 = o.x * Z1Z1;
        F S1 = y * o.z * Z2Z2, S2 = o.y * z * Z1Z1;
        if (U1 == U2) {
            if (S1 == S2) return dbl();
            return infinity();
        }
        F H = U2 - U1;
        F I = H.dbl().sqr();
        F J = H * I;
        F r = (S2 - S1).dbl();
        F V = U1 * I;
        F X3 = r.sqr() - J - V.dbl();
        F Y3 = r * (V - X3) - (S1 * J).dbl();
        F Z3 = ((z + o.z).sqr() - Z1Z1 - Z2Z2) * H;
        return {X3, Y3, Z3};
    }
    // madd-2007-bl
    JacT add_affine(const Affine& o) const {
        if (o.inf) return *this;
        if (is_inf()) return from_affine(o);
        F Z1Z1 = z.sqr();
        F U2 = o.x * Z1Z1;
        F S2 = o.y * z * Z1Z1;
        if (U2 == x) {
            if (S2 == y) return dbl();
            return infinity();
        }
        F H = U2 - x;
        F HH = H.sqr();
        F I = HH.dbl().dbl();
        F J = H * I;
        F r = (S2 - y).dbl();
        F V = x * I;
        F X3 = r.sqr() - J - V.dbl();
        F Y3 = r * (V - X3) - (y * J).dbl();
        F Z3 = (z + H).sqr() - Z1Z1 - HH;
        return {X3, Y3, Z3};
    }
    // scalar given as canonical little-endian limbs
    JacT mul(const uint64_t* e, int n) const {
        JacT r = infinity();
        for (int i = n * 64 - 1; i >= 0; i--) {
            r = r.dbl();
            if ((e[i / 64] >> (i % 64)) & 1) r = r.add(*this);
        }
        return r;
    }
    Affine to_affine() const {
        if (is_inf()) return Affine::infinity();
        F zi = z.inverse(), zi2 = zi.sqr();
        return {x * zi2, y * zi2 * zi, false};
    }
    bool operator==(const JacT& o) const {
        if (is_inf() || o.is_inf()) return is_inf() == o.is_inf();
        F Z1Z1 = z.sqr(), Z2Z2 = o.z.sqr();
        return x * Z2Z2 == o.x * Z1Z1 && y * o.z * Z2Z2 == o.y * z * Z1Z1;
    }
};

// naive reference: sum of double-and-add products
template <class J, class Fr>
static J msm_naive(const typename J::Affine* bases, const Fr* scalars, size_t n) {
    J acc = J::infinity();
    for (size_t i = 0; i < n; i++) {
        uint64_t e[Fr::N]; scalars[i].to_canonical(e);
        acc = acc.add(J::from_affine(bases[i]).mul(e, Fr::N));
    }
    return acc;
}

static inline int ark_window_size(size_t n) {
    if (n < 32) return 3;
    int lg = 63 - __builtin_clzll((unsigned long long)n);
    return lg * 69 / 100 + 2;
}

// ark-ec 0.4.2 msm_bigint restated; `threads` = worker threads over windows (arkworks: rayon over windows).
template <class J, class Fr>
static J msm_pippenger(const typename J::Affine* bases, const Fr* scalars, size_t n, int threads = 1) {
    typedef typename J::Affine A;
    const int c = ark_window_size(n);
    const int num_bits = Fr::K.bits;
    std::vector<uint64_t> big(n * Fr::N);
    for (size_t i = 0; i < n; i++) scalars[i].to_canonical(&big[i * Fr::N]);
    std::vector<int> starts;
    for (int w = 0; w < num_bits; w += c) starts.push_back(w);
    std::vector<J> sums(starts.size(), J::infinity());
    auto is_one = [&](const uint64_t* s) { if (s[0] != 1) return false; for (int k = 1; k < Fr::N; k++) if (s[k]) return false; return true; };
    auto is_zero = [&](const uint64_t* s) { for (int k = 0; k < Fr::N; k++) if (s[k]) return false; return true; };
    auto window = [&](size_t wi) {
        int w_start = starts[wi];
        J res = J::infinity();
        std::vector<J> buckets(((size_t)1 << c) - 1, J::infinity());
        for (size_t i = 0; i < n; i++) {
            const uint64_t* s = &big[i * Fr::N];
            if (is_zero(s)) continue;
            if (is_one(s)) { if (w_start == 0) res = res.add_affine(bases[i]); continue; }
            int limb = w_start / 64, off = w_start % 64;
            uint64_t d = s[limb] >> off;
            if (off + c > 64 && limb + 1 < Fr::N) d |= s[limb + 1] << (64 - off);
            d &= (((uint64_t)1 << c) - 1);
            if (d) buckets[d - 1] = buckets[d - 1].add_affine(bases[i]);
        }
        J running = J::infinity();
        for (size_t b = buckets.size(); b-- > 0;) { running = running.add(buckets[b]); res = res.add(running); }
        sums[wi] = res;
    };
    if (threads <= 1) { for (size_t wi = 0; wi < starts.size(); wi++) window(wi); }
    else {
        std::atomic<size_t> next(0);
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back([&] { for (;;) { size_t wi = next++; if (wi >= starts.size()) break; window(wi); } });
        for (auto& th : pool) th.join();
    }
    J total = J::infinity();
    for (size_t wi = starts.size(); wi-- > 1;) {
        total = total.add(sums[wi]);
        for (int k = 0; k < c; k++) total = total.dbl();
    }
    return sums[0].add(total);
}

}  // namespace orc
