// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// CPU restatement of the prime-field arithmetic the reference gets from the third-party crate
// ark-ff 0.4.2 (`/root/reference/Cargo.toml:36`; sources are NOT in /root/reference, so the published
// algorithm is restated): Montgomery-form Fp with R = 2^(64*N), fully reduced representatives.
// Reference call sites that fix the conventions used here:
//   * in-memory element = N x u64 little-endian limbs in Montgomery form
//     (`co-circom/circom-types/src/traits.rs:57-67`  montgomery_bigint_from_reader / new_unchecked)
//   * zkey section-4 coefficient decode value*R^2 -> value  (`traits.rs:65-67`)
//   * wtns values are canonical little-endian  (`traits.rs:50-54`, `witness.rs:51-91`)
//
// Deliberately written differently from the product's field code (separate-operand-scanning product +
// word-by-word REDC on 64-bit limbs) so that GPU-vs-oracle agreement is agreement of two implementations.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include <algorithm>

namespace orc {

typedef unsigned __int128 u128;

template <int N>
struct FieldConsts {
    uint64_t p[N];    // modulus
    uint64_t r1[N];   // R mod p   (Montgomery one)
    uint64_t r2[N];   // R^2 mod p
    uint64_t inv;     // -p^{-1} mod 2^64
    int bits;         // bit length of p
    bool ready = false;
};

// ---- raw limb helpers -------------------------------------------------------------------------
template <int N>
static inline uint64_t raw_add(uint64_t* r, const uint64_t* a, const uint64_t* b) {
    u128 c = 0;
    for (int i = 0; i < N; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
template <int N>
static inline uint64_t raw_sub(uint64_t* r, const uint64_t* a, const uint64_t* b) {
    uint64_t borrow = 0;
    for (int i = 0; i < N; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
template <int N>
static inline int raw_cmp(const uint64_t* a, const uint64_t* b) {
    for (int i = N - 1; i >= 0; i--) { if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1; }
    return 0;
}
static inline std::vector<uint64_t> limbs_from_hex(const char* hex, int n) {
    std::vector<uint64_t> out(n, 0);
    std::string s(hex);
    if (s.size() >= 2 && s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) s = s.substr(2);
    int bit = 0;
    for (int i = (int)s.size() - 1; i >= 0; i--) {
        char c = s[i]; int d;
        if (c >= '0' && c <= '9') d = c - '0';
        else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
        else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
        else continue;
        if (bit / 64 < n) out[bit / 64] |= (uint64_t)d << (bit % 64);
        bit += 4;
    }
    return out;
}

// ---- Fp ----------------------------------------------------------------------------------------
template <int N_, int TAG>
struct Fp {
    static constexpr int N = N_;
    uint64_t v[N];   // Montgomery representation, < p

    static inline FieldConsts<N> K;

    static void init(const char* modulus_hex) {
        auto m = limbs_from_hex(modulus_hex, N);
        for (int i = 0; i < N; i++) K.p[i] = m[i];
        // -p^{-1} mod 2^64 by Newton iteration
        uint64_t x = 1;
        for (int i = 0; i < 6; i++) x *= 2 - K.p[0] * x;
        K.inv = (uint64_t)0 - x;
        int bits = 64 * N;
        while (bits > 0 && !((K.p[(bits - 1) / 64] >> ((bits - 1) % 64)) & 1)) bits--;
        K.bits = bits;
        // R mod p, R^2 mod p by repeated doubling of 1
        uint64_t t[N] = {0}; t[0] = 1;
        auto dbl = [&](uint64_t* a) {
            uint64_t c = raw_add<N>(a, a, a);
            if (c || raw_cmp<N>(a, K.p) >= 0) raw_sub<N>(a, a, K.p);
        };
        for (int i = 0; i < 64 * N; i++) dbl(t);
        memcpy(K.r1, t, sizeof t);
        for (int i = 0; i < 64 * N; i++) dbl(t);
        memcpy(K.r2, t, sizeof t);
        K.ready = true;
    }

    static Fp zero() { Fp r; memset(r.v, 0, sizeof r.v); return r; }
    static Fp one() { Fp r; memcpy(r.v, K.r1, sizeof r.v); return r; }
    static Fp from_mont_limbs(const uint64_t* l) { Fp r; memcpy(r.v, l, sizeof r.v); return r; }
    // canonical integer (may be >= p: reduced mod p first, like from_le_bytes_mod_order for < 2^(64N))
    static Fp from_canonical(const uint64_t* l) {
        Fp t; memcpy(t.v, l, sizeof t.v);
        while (raw_cmp<N>(t.v, K.p) >= 0) raw_sub<N>(t.v, t.v, K.p);
        Fp r2 = from_mont_limbs(K.r2);
        return t * r2;
    }
    static Fp from_u64(uint64_t x) { uint64_t l[N] = {0}; l[0] = x; return from_canonical(l); }
    static Fp from_dec(const std::string& s) {
        Fp acc = zero(), ten = from_u64(10);
        bool neg = false;
        for (char c : s) {
            if (c == '-') { neg = true; continue; }
            if (c < '0' || c > '9') continue;
            acc = acc * ten + from_u64((uint64_t)(c - '0'));
        }
        return neg ? -acc : acc;
    }
    void to_canonical(uint64_t* out) const {
        Fp o; memset(o.v, 0, sizeof o.v); o.v[0] = 1;   // raw 1 (not Montgomery)
        Fp r = (*this) * o;
        memcpy(out, r.v, sizeof r.v);
    }
    std::string to_dec() const {
        uint64_t c[N]; to_canonical(c);
        std::string out;
        bool nz = true;
        while (nz) {
            u128 rem = 0; nz = false;
            for (int i = N - 1; i >= 0; i--) {
                u128 cur = (rem << 64) | c[i];
                c[i] = (uint64_t)(cur / 10); rem = cur % 10;
                if (c[i]) nz = true;
            }
            out.push_back((char)('0' + (int)rem));
        }
        std::reverse(out.begin(), out.end());
        return out;
    }
    bool is_zero() const { for (int i = 0; i < N; i++) if (v[i]) return false; return true; }
    bool operator==(const Fp& o) const { return memcmp(v, o.v, sizeof v) == 0; }
    bool operator!=(const Fp& o) const { return !(*this == o); }

    Fp operator+(const Fp& o) const {
        Fp r; uint64_t c = raw_add<N>(r.v, v, o.v);
        if (c || raw_cmp<N>(r.v, K.p) >= 0) raw_sub<N>(r.v, r.v, K.p);
        return r;
    }
    Fp operator-(const Fp& o) const {
        Fp r; uint64_t b = raw_sub<N>(r.v, v, o.v);
        if (b) raw_add<N>(r.v, r.v, K.p);
        return r;
    }
    Fp operator-() const { if (is_zero()) return *this; Fp r; raw_sub<N>(r.v, K.p, v); return r; }
    Fp dbl() const { return *this + *this; }

    // Montgomery product: full 2N-limb schoolbook product, then N rounds of word REDC (loops fully unrolled by the
    // compiler for the fixed N; p < 2^(64N-1) so the running value never needs more than 2N limbs + the final compare).
    Fp operator*(const Fp& o) const {
        uint64_t t[2 * N];
#pragma GCC unroll 16
        for (int j = 0; j < 2 * N; j++) t[j] = 0;
#pragma GCC unroll 16
        for (int i = 0; i < N; i++) {
            uint64_t c = 0;
#pragma GCC unroll 16
            for (int j = 0; j < N; j++) {
                u128 x = (u128)v[i] * o.v[j] + t[i + j] + c;
                t[i + j] = (uint64_t)x; c = (uint64_t)(x >> 64);
            }
            t[i + N] = c;
        }
        uint64_t carry = 0;
#pragma GCC unroll 16
        for (int i = 0; i < N; i++) {
            uint64_t m = t[i] * K.inv;
            uint64_t c = 0;
#pragma GCC unroll 16
            for (int j = 0; j < N; j++) {
                u128 x = (u128)m * K.p[j] + t[i + j] + c;
                t[i + j] = (uint64_t)x; c = (uint64_t)(x >> 64);
            }
            u128 x = (u128)t[i + N] + c + carry;
            t[i + N] = (uint64_t)x; carry = (uint64_t)(x >> 64);
        }
        Fp r;
#pragma GCC unroll 16
        for (int j = 0; j < N; j++) r.v[j] = t[N + j];
        if (carry || raw_cmp<N>(r.v, K.p) >= 0) raw_sub<N>(r.v, r.v, K.p);
        return r;
    }
    Fp sqr() const { return (*this) * (*this); }
    Fp& operator+=(const Fp& o) { *this = *this + o; return *this; }
    Fp& operator-=(const Fp& o) { *this = *this - o; return *this; }
    Fp& operator*=(const Fp& o) { *this = *this * o; return *this; }

    // exponent given as little-endian u64 limbs
    Fp pow(const uint64_t* e, int n) const {
        Fp r = one();
        for (int i = n * 64 - 1; i >= 0; i--) {
            r = r.sqr();
            if ((e[i / 64] >> (i % 64)) & 1) r = r * (*this);
        }
        return r;
    }
    Fp inverse() const {   // Fermat: a^(p-2); inverse of 0 is 0
        uint64_t e[N]; uint64_t two[N] = {0}; two[0] = 2;
        raw_sub<N>(e, K.p, two);
        return pow(e, N);
    }
    // Legendre symbol via a^((p-1)/2): returns 1, -1 or 0
    int legendre() const {
        uint64_t e[N]; uint64_t o[N] = {0}; o[0] = 1;
        raw_sub<N>(e, K.p, o);
        for (int i = 0; i < N; i++) e[i] = (e[i] >> 1) | (i + 1 < N ? e[i + 1] << 63 : 0);
        Fp r = pow(e, N);
        if (r.is_zero()) return 0;
        return r == one() ? 1 : -1;
    }
};

// batch inversion (Montgomery trick); zeros stay zero
template <class F>
static inline void batch_inverse(F* a, size_t n) {
    std::vector<F> pre(n);
    F acc = F::one();
    for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!a[i].is_zero()) acc = acc * a[i]; }
    F inv = acc.inverse();
    for (size_t i = n; i-- > 0;) {
        if (a[i].is_zero()) continue;
        F t = inv * pre[i];
        inv = inv * a[i];
        a[i] = t;
    }
}

// ---- Fp2 = Fp[u]/(u^2+1)  (both BN254 and BLS12-381 use non-residue -1) --------------------------
template <class F>
struct Fp2T {
    typedef F Base;
    F c0, c1;
    static Fp2T zero() { return {F::zero(), F::zero()}; }
    static Fp2T one() { return {F::one(), F::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fp2T& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Fp2T& o) const { return !(*this == o); }
    Fp2T operator+(const Fp2T& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fp2T operator-(const Fp2T& o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fp2T operator-() const { return {-c0, -c1}; }
    Fp2T dbl() const { return {c0.dbl(), c1.dbl()}; }
    Fp2T operator*(const Fp2T& o) const {
        F a = c0 * o.c0, b = c1 * o.c1;
        F c = (c0 + c1) * (o.c0 + o.c1);
        return {a - b, c - a - b};
    }
    Fp2T sqr() const { return (*this) * (*this); }
    Fp2T mul_base(const F& s) const { return {c0 * s, c1 * s}; }
    Fp2T conj() const { return {c0, -c1}; }
    Fp2T inverse() const {
        F n = (c0.sqr() + c1.sqr()).inverse();
        return {c0 * n, -(c1 * n)};
    }
    Fp2T& operator+=(const Fp2T& o) { *this = *this + o; return *this; }
    Fp2T& operator-=(const Fp2T& o) { *this = *this - o; return *this; }
    Fp2T& operator*=(const Fp2T& o) { *this = *this * o; return *this; }
};

}  // namespace orc
