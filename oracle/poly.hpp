// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// Radix-2 NTT restating what the reference obtains from ark-poly 0.4.2 `Radix2EvaluationDomain::{fft,ifft}_in_place`
// (`/root/reference/Cargo.toml:37`, not vendored; call sites `/root/reference/mpc-core/src/protocols/rep3.rs:895-896,918-919`,
// `plain.rs:375-406`), with the generator injected by the caller exactly like
// `/root/reference/co-circom/co-groth16/src/groth16.rs:57-77` overrides `domain.group_gen`:
//   forward : X[k] = sum_j x[j] w^(jk)            natural order in and out   (ark: DIF "io" pass + bit-reversal)
//   inverse : x[j] = m^-1 sum_k X[k] w^(-jk)      natural order in and out   (ark: bit-reversal + DIT "oi" pass + scale)
// and the snarkjs root-of-unity table of `/root/reference/co-circom/co-circom-snarks/src/lib.rs:208-221`.
#pragma once
#include "ff.hpp"

namespace orc {

static inline size_t bitrev(size_t x, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
static inline int log2_exact(size_t n) { int l = 0; while (((size_t)1 << l) < n) l++; return l; }

template <class F>
static void bitrev_permute(F* a, size_t n) {
    int lg = log2_exact(n);
    for (size_t i = 0; i < n; i++) { size_t j = bitrev(i, lg); if (i < j) std::swap(a[i], a[j]); }
}

// forward NTT, generator w of order n: decimation-in-frequency then bit reversal
template <class F>
static void ntt_forward(F* a, size_t n, const F& w) {
    if (n <= 1) return;
    std::vector<F> tw(n / 2);
    tw[0] = F::one();
    for (size_t i = 1; i < n / 2; i++) tw[i] = tw[i - 1] * w;
    for (size_t half = n / 2, step = 1; half >= 1; half >>= 1, step <<= 1) {
        for (size_t blk = 0; blk < n; blk += 2 * half)
            for (size_t j = 0; j < half; j++) {
                F u = a[blk + j], v = a[blk + j + half];
                a[blk + j] = u + v;
                a[blk + j + half] = (u - v) * tw[j * step];
            }
    }
    bitrev_permute(a, n);
}

// inverse NTT: bit reversal, decimation-in-time with w^-1, scale by n^-1
template <class F>
static void ntt_inverse(F* a, size_t n, const F& w) {
    if (n <= 1) return;
    F wi = w.inverse();
    std::vector<F> tw(n / 2);
    tw[0] = F::one();
    for (size_t i = 1; i < n / 2; i++) tw[i] = tw[i - 1] * wi;
    bitrev_permute(a, n);
    for (size_t half = 1, step = n / 2; half < n; half <<= 1, step >>= 1) {
        for (size_t blk = 0; blk < n; blk += 2 * half)
            for (size_t j = 0; j < half; j++) {
                F u = a[blk + j], v = a[blk + j + half] * tw[j * step];
                a[blk + j] = u + v;
                a[blk + j + half] = u - v;
            }
    }
    F ninv = F::from_u64((uint64_t)n).inverse();
    for (size_t i = 0; i < n; i++) a[i] = a[i] * ninv;
}

// O(n^2) definition, used by tests to pin the fast transforms
template <class F>
static std::vector<F> dft_naive(const F* a, size_t n, const F& w) {
    std::vector<F> out(n);
    F wk = F::one();
    for (size_t k = 0; k < n; k++) {
        F acc = F::zero(), x = F::one();
        for (size_t j = 0; j < n; j++) { acc = acc + a[j] * x; x = x * wk; }
        out[k] = acc; wk = wk * w;
    }
    return out;
}

// `/root/reference/co-circom/co-circom-snarks/src/lib.rs:208-221`: q = smallest quadratic non-residue,
// z = q^T (T = odd part of p-1), roots = [z, z^2, z^4, ...] reversed => roots[i] is a primitive 2^i-th root.
template <class F>
struct SnarkjsRoots {
    F q;
    std::vector<F> roots;
    int two_adicity;
};
template <class F>
static SnarkjsRoots<F> roots_of_unity() {
    SnarkjsRoots<F> out;
    uint64_t t[F::N]; uint64_t one[F::N] = {0}; one[0] = 1;
    raw_sub<F::N>(t, F::K.p, one);
    int s = 0;
    while (!(t[0] & 1)) {
        for (int i = 0; i < F::N; i++) t[i] = (t[i] >> 1) | (i + 1 < F::N ? t[i + 1] << 63 : 0);
        s++;
    }
    out.two_adicity = s;
    F q = F::one();
    while (q.legendre() != -1) q = q + F::one();
    out.q = q;
    std::vector<F> r(s + 1);
    r[0] = q.pow(t, F::N);
    for (int i = 1; i <= s; i++) r[i] = r[i - 1].sqr();
    std::reverse(r.begin(), r.end());
    out.roots = r;
    return out;
}

}  // namespace orc
