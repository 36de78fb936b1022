// Round 6: the batched-affine question for G2 (rounds 3-4 closed it for G1: profiles/r04_microbench_affine2.txt).  In Fq2 the mixed XYZZ
// addition costs 8 M + 2 S = 4 536 multiply-adds and the shipped kernel sustains 4.67 G additions/s (54.5 M in 11.67 ms); an affine addition with the
// inverse of the lane's running product given costs 5 M + 1 S = 2 754, and a point is twice as many bytes as in G1 — so the HBM ceiling that stopped G1
// (360 B per addition at the copy rate: 14.3 G/s against XYZZ's 17.5) is 720 B -> ~7.2 G/s here, ABOVE what the kernel does today.  This file measures
// section B of microbench_affine2.hip on the product's Fq2 lazy core (L29x2): B = 16 .. 128 additions per lane, operands, running products and
// results staged through HBM in limb form (coalesced planes), inverse GIVEN / one inversion per wave (Fq2 inverse = conjugate over the norm: two
// squarings, one binary-GCD inversion in Fq, two products).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../include scripts/microbench_affine_g2.hip -o scripts/_build/microbench_affine_g2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../collaborative-circom_amd/csrc/common.hpp"
#include "../collaborative-circom_amd/csrc/msm_kernels.hpp"
using namespace cg;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
typedef L29<Bn254Fq> L;
typedef Bn254Fq::Params P;
constexpr int NL = L::NL;

__device__ __forceinline__ uint32_t hash32(uint32_t v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }
__device__ __forceinline__ L gen(uint32_t key) {
    L r;
    _Pragma("unroll") for (int k = 0; k < NL; k++) r.l[k] = (int32_t)(hash32(key * 16u + k) & L::MASK);
    r.l[NL - 1] &= 0x3fffff;
    return r;
}

// ---- A. inversions -------------------------------------------------------------------------------------------------------------
__device__ __noinline__ L fermat_inverse(const L& a) {
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = P::P[i];
    e[0] -= 2;
    L r = a; bool started = false;
    for (int i = 255; i >= 0; i--) {
        const bool bit = (e[i >> 5] >> (i & 31)) & 1u;
        if (!started) { started = bit; continue; }
        r = L::sqr(r);
        if (bit) r = L::mul(r, a);
    }
    return r;
}
// 8 x 32-bit helpers for the binary GCD (plain words; one lane works, so branches cost nothing)
struct W8 { uint32_t w[8]; };
__device__ __forceinline__ bool w8_is_one(const W8& a) { uint32_t o = a.w[0] ^ 1u; for (int i = 1; i < 8; i++) o |= a.w[i]; return o == 0; }
__device__ __forceinline__ bool w8_geq(const W8& a, const W8& b) { for (int i = 7; i >= 0; i--) { if (a.w[i] != b.w[i]) return a.w[i] > b.w[i]; } return true; }
__device__ __forceinline__ uint32_t w8_sub(W8& a, const W8& b) { uint64_t br = 0; for (int i = 0; i < 8; i++) { const uint64_t t = (uint64_t)a.w[i] - b.w[i] - br; a.w[i] = (uint32_t)t; br = (t >> 63) & 1; } return (uint32_t)br; }
__device__ __forceinline__ uint32_t w8_add(W8& a, const W8& b) { uint64_t c = 0; for (int i = 0; i < 8; i++) { c += (uint64_t)a.w[i] + b.w[i]; a.w[i] = (uint32_t)c; c >>= 32; } return (uint32_t)c; }
__device__ __forceinline__ void w8_shr1(W8& a, uint32_t top) { for (int i = 0; i < 7; i++) a.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 31); a.w[7] = (a.w[7] >> 1) | (top << 31); }
// a^-1 mod p for a canonical value 0 < a < p (HAC 14.61 shape: u, v shrink by halving and subtraction; x1, x2 follow modulo p)
__device__ __noinline__ W8 binary_gcd_inverse(const W8& a) {
    W8 pm; for (int i = 0; i < 8; i++) pm.w[i] = P::P[i];
    W8 u = a, v = pm, x1, x2;
    for (int i = 0; i < 8; i++) { x1.w[i] = i == 0; x2.w[i] = 0; }
    while (!w8_is_one(u) && !w8_is_one(v)) {
        while (!(u.w[0] & 1u)) { w8_shr1(u, 0); uint32_t c = 0; if (x1.w[0] & 1u) c = w8_add(x1, pm); w8_shr1(x1, c); }
        while (!(v.w[0] & 1u)) { w8_shr1(v, 0); uint32_t c = 0; if (x2.w[0] & 1u) c = w8_add(x2, pm); w8_shr1(x2, c); }
        if (w8_geq(u, v)) { w8_sub(u, v); if (w8_sub(x1, x2)) w8_add(x1, pm); }
        else { w8_sub(v, u); if (w8_sub(x2, x1)) w8_add(x2, pm); }
    }
    return w8_is_one(u) ? x1 : x2;
}
// inverse in the core's 2^261 Montgomery domain through the word-level GCD: x (value a 2^261) -> canonical words of the residue a 2^256
// (to_fp), word inverse (a 2^256)^-1, back into limbs and two products by constants bring it to a^-1 2^261
__device__ __forceinline__ L gcd_inverse(const L& x, const L& fix) {
    const Bn254Fq f = L::to_fp(x);                               // a * 2^256 mod p, canonical
    W8 a; for (int i = 0; i < 8; i++) a.w[i] = f.v[i];
    const W8 iv = binary_gcd_inverse(a);                         // a^-1 2^-256
    Bn254Fq g; for (int i = 0; i < 8; i++) g.v[i] = iv.w[i];
    return L::mul(L::template unpack<0>(g), fix);                // (a^-1 2^-256) * fix / 2^261 with fix = 2^(256 + 2*261): a^-1 2^261
}
// 2^e mod p as an integer in limbs (computed once by one lane of k_fix and read back by the kernels)
__global__ void k_fix(int32_t* out, int e) {
    W8 x; for (int i = 0; i < 8; i++) x.w[i] = i == 0;
    W8 pm; for (int i = 0; i < 8; i++) pm.w[i] = P::P[i];
    for (int i = 0; i < e; i++) { const uint32_t top = x.w[7] >> 31; for (int k = 7; k > 0; k--) x.w[k] = (x.w[k] << 1) | (x.w[k - 1] >> 31); x.w[0] <<= 1; if (top || w8_geq(x, pm)) w8_sub(x, pm); }
    Bn254Fq g; for (int i = 0; i < 8; i++) g.v[i] = x.w[i];
    const L r = L::template unpack<0>(g);
    for (int k = 0; k < NL; k++) out[k] = r.l[k];
}
__device__ __forceinline__ L ld_fix(const int32_t* f) { L r; _Pragma("unroll") for (int k = 0; k < NL; k++) r.l[k] = f[k]; return r; }

__global__ void __launch_bounds__(256) k_inv_single_lane(uint32_t* out, uint32_t seed, int iters, int kind, uint32_t* check, const int32_t* fixp) {
    const uint32_t lane = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x & 63u) return;                               // one active lane per wave
    L a = gen(seed ^ lane);
    const L fix = ld_fix(fixp);
    uint32_t bad = 0;
    for (int it = 0; it < iters; it++) {
        const L inv = kind == 0 ? fermat_inverse(a) : gcd_inverse(a, fix);
        if (it == 0) { const Bn254Fq one = L::to_fp(L::mul(a, inv)); bad |= !(one == Bn254Fq::one()); }
        a = (inv + a).norm();
    }
    uint32_t x = 0; for (int k = 0; k < NL; k++) x ^= (uint32_t)a.l[k];
    out[lane] = x;
    if (bad) atomicOr(check, 1u << kind);
}
// reference for "product times": a chain of products on one lane per wave, same launch shape
__global__ void __launch_bounds__(256) k_mul_single_lane(uint32_t* out, uint32_t seed, int iters) {
    const uint32_t lane = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x & 63u) return;
    L a = gen(seed ^ lane), b = gen(seed + 77u + lane);
    for (int it = 0; it < iters; it++) { a = L::mul(a, b); b = L::mul(b, a); }
    uint32_t x = 0; for (int k = 0; k < NL; k++) x ^= (uint32_t)(a.l[k] ^ b.l[k]);
    out[lane] = x;
}


// ---- B. batches staged through HBM, Fq2 -------------------------------------------------------------------------------------------
typedef L29x2<Fp2<Bn254Fq>> L2;
__device__ __forceinline__ L ld_plane(const int32_t* base, size_t lanes, uint32_t q) { L r; _Pragma("unroll") for (int k = 0; k < NL; k++) r.l[k] = base[(size_t)k * lanes + q]; return r; }
__device__ __forceinline__ void st_plane(int32_t* base, size_t lanes, uint32_t q, const L& v) { _Pragma("unroll") for (int k = 0; k < NL; k++) base[(size_t)k * lanes + q] = v.l[k]; }
// an Fq2 value = two consecutive limb planes (c0 then c1)
__device__ __forceinline__ L2 ld2(const int32_t* base, size_t lanes, uint32_t q) { return {ld_plane(base, lanes, q), ld_plane(base + (size_t)NL * lanes, lanes, q)}; }
__device__ __forceinline__ void st2(int32_t* base, size_t lanes, uint32_t q, const L2& v) { st_plane(base, lanes, q, v.c0); st_plane(base + (size_t)NL * lanes, lanes, q, v.c1); }
__global__ void __launch_bounds__(256) k_fill(int32_t* pts, size_t n, uint32_t seed, size_t lanes) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        pts[i] = (int32_t)(hash32((uint32_t)i * 2654435761u + seed) & ((i / lanes) % NL == NL - 1 ? 0x1fffffu : 0x0fffffffu));
}
__device__ __forceinline__ L shfl_L(const L& v, int src) { L r; _Pragma("unroll") for (int k = 0; k < NL; k++) r.l[k] = __shfl(v.l[k], src, 64); return r; }
__device__ __forceinline__ L2 shfl_L2(const L2& v, int src) { return {shfl_L(v.c0, src), shfl_L(v.c1, src)}; }
__device__ __forceinline__ L2 gen2(uint32_t key) { return {gen(key), gen(key ^ 0x9e3779b9u)}; }
// 1 / a in Fq2 = conj(a) / (a0^2 + a1^2)
__device__ __forceinline__ L2 inverse2(const L2& a, const L& fix) {
    const L n = (L::sqr(a.c0) + L::sqr(a.c1)).norm();
    const L ni = gcd_inverse(n, fix);
    return {L::mul(a.c0, ni), L::mul(a.c1.neg().norm(), ni)};
}

template <int MODE /* 0 = inverse given, 1 = one inversion per wave */>
__global__ void __launch_bounds__(256, 2) k_affine_hbm_g2(const int32_t* __restrict__ pts, int32_t* __restrict__ pre, int32_t* __restrict__ outp, int B, size_t lanes, uint32_t* check, const int32_t* fixp) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x, wl = threadIdx.x & 63u;
    const size_t es = (size_t)2 * NL * lanes;                                     // one Fq2 element (two planes of NL limbs)
    const size_t cs = (size_t)B * es, js = es;                                    // coordinate stride, addition stride
    L2 run;
    for (int j = 0; j < B; j++) {
        const L2 x1 = ld2(pts + 0 * cs + j * js, lanes, q), x2 = ld2(pts + 2 * cs + j * js, lanes, q);
        const L2 dx = x2 - x1;
        run = j ? L2::mul(run, dx) : dx.norm();
        st2(pre + j * js, lanes, q, run);
    }
    L2 I;
    if constexpr (MODE == 0) I = gen2(q);
    else {
        L2 pf = run, sf = run;
        _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) {
            const L2 up = shfl_L2(pf, (int)wl - d), dn = shfl_L2(sf, (int)wl + d);
            if ((int)wl - d >= 0) pf = L2::mul(pf, up);
            if ((int)wl + d < 64) sf = L2::mul(sf, dn);
        }
        const L2 pe = shfl_L2(pf, (int)wl - 1), se = shfl_L2(sf, (int)wl + 1);
        L2 inv_total;
        if (wl == 63) inv_total = inverse2(pf, ld_fix(fixp));
        inv_total = shfl_L2(inv_total, 63);
        const L2 others = wl == 0 ? se : (wl == 63 ? pe : L2::mul(pe, se));
        I = L2::mul(inv_total, others);
        if (blockIdx.x == 0 && (wl == 5 || wl == 0 || wl == 63)) {
            const L2 one = L2::mul(run, I);
            const Fp2<Bn254Fq> o = L2::to_fp(one);
            if (!(o.c0 == Bn254Fq::one()) || !(o.c1 == Bn254Fq::zero())) atomicOr(check, 4u);
        }
    }
    for (int j = B - 1; j >= 0; j--) {
        const L2 x1 = ld2(pts + 0 * cs + j * js, lanes, q), y1 = ld2(pts + 1 * cs + j * js, lanes, q);
        const L2 x2 = ld2(pts + 2 * cs + j * js, lanes, q), y2 = ld2(pts + 3 * cs + j * js, lanes, q);
        L2 inv = I;
        if (j > 0) { const L2 pj = ld2(pre + (size_t)(j - 1) * js, lanes, q); inv = L2::mul(I, pj); I = L2::mul(I, (x2 - x1).norm()); }
        const L2 lam = L2::mul((y2 - y1).norm(), inv);
        const L2 x3 = (L2::sqr(lam) - x1 - x2).norm();
        const L2 y3 = (L2::mul(lam, (x1 - x3).norm()) - y1).norm();
        st2(outp + 0 * cs + j * js, lanes, q, x3);
        st2(outp + 1 * cs + j * js, lanes, q, y3);
    }
}

int main() {
    uint32_t* chk;
    const int WG = 1024;                                           // 2 full residency rounds of 2 workgroups per CU
    const size_t lanes = (size_t)WG * 256;
    CHK(hipMalloc(&chk, 4)); CHK(hipMemset(chk, 0, 4));
    int32_t* fixp; CHK(hipMalloc(&fixp, NL * 4));
    hipLaunchKernelGGL(k_fix, dim3(1), dim3(1), 0, 0, fixp, 256 + 2 * 261); CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto ms_of = [&](auto launch) -> float { float best = 1e30f; for (int rep = 0; rep < 3; rep++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; } return best; };
    printf("== batched affine additions in G2 (Fq2 on 2 x 9 x 29-bit lazy limbs), operands / running products / results staged through HBM (720 B per addition), %d workgroups x 256 lanes\n", WG);
    for (int B : {16, 32, 64, 128}) {
        int32_t *pts, *pre, *outp;
        const size_t plane = (size_t)B * 2 * NL * lanes;
        CHK(hipMalloc(&pts, 4 * plane * 4)); CHK(hipMalloc(&pre, plane * 4)); CHK(hipMalloc(&outp, 2 * plane * 4));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, pts, 4 * plane, 11u, lanes);
        CHK(hipDeviceSynchronize());
        const double adds = (double)lanes * B, bytes = adds * 720.0;
        const float t0 = ms_of([&] { hipLaunchKernelGGL((k_affine_hbm_g2<0>), dim3(WG), dim3(256), 0, 0, pts, pre, outp, B, lanes, chk, fixp); });
        const float t1 = ms_of([&] { hipLaunchKernelGGL((k_affine_hbm_g2<1>), dim3(WG), dim3(256), 0, 0, pts, pre, outp, B, lanes, chk, fixp); });
        printf("B = %3d additions per lane (%5.0f per inversion): inverse given %6.2f G add/s (%5.2f TB/s moved, %.2f ms per 54.5 M) | one inversion per wave %6.2f G add/s (%.2f ms per 54.5 M)\n",
               B, 64.0 * B, adds / (t0 * 1e-3) / 1e9, bytes / (t0 * 1e-3) / 1e12, 54.5e6 / (adds / (t0 * 1e-3)) * 1e3, adds / (t1 * 1e-3) / 1e9, 54.5e6 / (adds / (t1 * 1e-3)) * 1e3);
        CHK(hipFree(pts)); CHK(hipFree(pre)); CHK(hipFree(outp));
    }
    uint32_t bad; CHK(hipMemcpy(&bad, chk, 4, hipMemcpyDeviceToHost));
    printf("self-check (lane product * its inverse == 1 through the wave scans): %s (flags %u)\n", bad ? "FAILED" : "ok", bad);
    printf("reference: XYZZ mixed addition in G2 inside k_msm_accumulate_pf<G2>: 4.67 G additions/s (54.5 M in 11.67 ms)\n");
    return bad ? 2 : 0;
}
