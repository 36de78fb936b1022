// Pointwise secret-shared field arithmetic and the sparse constraint evaluation of the co-groth16 witness map.
// HBM-bound kernels: one 32-byte element per lane per iteration, loaded as two 16-byte halves (global_load_dwordx4),
// grid-stride over ~2048 workgroups.  Reference semantics cited per kernel.
#pragma once
#include "field.hpp"

namespace cg {

template <class F>
__device__ __forceinline__ F ld_fp(const F* p) {
    F r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    _Pragma("unroll") for (int i = 0; i < F::N / 4; i++) {
        uint4 w = q[i];
        r.v[4 * i] = w.x; r.v[4 * i + 1] = w.y; r.v[4 * i + 2] = w.z; r.v[4 * i + 3] = w.w;
    }
    return r;
}
template <class F>
__device__ __forceinline__ void st_fp(F* p, const F& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    _Pragma("unroll") for (int i = 0; i < F::N / 4; i++) q[i] = make_uint4(r.v[4 * i], r.v[4 * i + 1], r.v[4 * i + 2], r.v[4 * i + 3]);
}

// op: 0 add, 1 sub, 2 mul   (plain.rs:215-224 add_vec / mul_vec; rep3.rs:672-679 sub_assign_vec per component)
template <class F, int OP>
__global__ void __launch_bounds__(256) k_vec_binary(F* __restrict__ out, const F* __restrict__ a, const F* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        F x = ld_fp(a + i), y = ld_fp(b + i);
        st_fp(out + i, OP == 0 ? x + y : OP == 1 ? x - y : x * y);
    }
}

// REP3 local product (rep3.rs:656-660, fieldshare.rs:161-168): out = aa*ba + aa*bb + ab*ba (+ mask)
// computed as aa*(ba+bb) + ab*ba: two Montgomery products instead of three.
template <class F>
__global__ void __launch_bounds__(256) k_rep3_mul_local(F* __restrict__ out, const F* __restrict__ aa, const F* __restrict__ ab,
                                                        const F* __restrict__ ba, const F* __restrict__ bb, const F* __restrict__ mask, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        F xa = ld_fp(aa + i), xb = ld_fp(ab + i), ya = ld_fp(ba + i), yb = ld_fp(bb + i);
        F r = xa * (ya + yb) + xb * ya;
        if (mask) r = r + ld_fp(mask + i);
        st_fp(out + i, r);
    }
}

// elements whose limbs are not below the modulus (what a deserialiser rejects): counted into *n_bad.  32 B read per element.
template <class F>
__global__ void __launch_bounds__(256) k_vec_count_noncanonical(const F* __restrict__ v, size_t n, unsigned long long* __restrict__ n_bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const F a = ld_fp(v + i);
        bool below = false, decided = false;
        _Pragma("unroll") for (int j = F::N - 1; j >= 0; j--) {
            if (!decided && a.v[j] != F::Params::P[j]) { below = a.v[j] < F::Params::P[j]; decided = true; }
        }
        if (!below) atomicAdd(n_bad, 1ull);
    }
}

// v[i] *= c * g^i  (rep3.rs:681-688 / plain.rs:226-234).  g^i = hi[i >> LOG_LO] * lo[i & (2^LOG_LO - 1)] from two small
// tables (lo[j] = c*g^j, hi[j] = g^(j << LOG_LO)) that stay in L2: no per-element power chain, no m-entry table in HBM.
template <class F>
__global__ void __launch_bounds__(256) k_distribute_powers(F* __restrict__ v, size_t n, const F* __restrict__ lo, const F* __restrict__ hi, int log_lo) {
    const size_t mask = ((size_t)1 << log_lo) - 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        F w = ld_fp(lo + (i & mask)) * ld_fp(hi + (i >> log_lo));
        st_fp(v + i, ld_fp(v + i) * w);
    }
}

// Constraint evaluation (groth16.rs:159-166 calling rep3.rs:690-708 / plain.rs:243-258) as a CSR sparse mat-vec,
// one row per lane.  Signal index < n_inputs -> public input (REP3: added to component `a` by party 0, to `b` by party 1,
// dropped by party 2 — rep3.rs:600-608); otherwise the shared witness at index - n_inputs.
// party: -1 = single-component driver (plain / Shamir: public inputs are added as they are), 0..2 = REP3 party id.
template <class F>
__global__ void __launch_bounds__(256) k_spmv_csr(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const F* __restrict__ coeff,
                                                  size_t n_rows, const F* __restrict__ pub, uint32_t n_inputs, int party,
                                                  const F* __restrict__ wit_a, const F* __restrict__ wit_b, F* __restrict__ out_a, F* __restrict__ out_b) {
    for (size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x; row < n_rows; row += (size_t)gridDim.x * blockDim.x) {
        F acc_a = F::zero(), acc_b = F::zero();
        const uint32_t e = row_ptr[row + 1];
        for (uint32_t k = row_ptr[row]; k < e; k++) {
            const uint32_t idx = col[k];
            const F c = ld_fp(coeff + k);
            if (idx < n_inputs) {
                F t = c * ld_fp(pub + idx);
                if (party <= 0) acc_a = acc_a + t;
                else if (party == 1) acc_b = acc_b + t;
            } else {
                acc_a = acc_a + c * ld_fp(wit_a + (idx - n_inputs));
                if (wit_b) acc_b = acc_b + c * ld_fp(wit_b + (idx - n_inputs));
            }
        }
        st_fp(out_a + row, acc_a);
        if (out_b) st_fp(out_b + row, acc_b);
    }
}

// The same rows with one 64-lane WAVE per row (small circuits: a few hundred rows cannot fill the chip with a lane each, and the launch then
// lasts as long as its longest row — the Poseidon fixture's 213 rows took 107 us, one row of ~100 terms at ~1 us per term).  Lanes stride over
// the row's terms and their partial sums are folded across the wave; field addition is exact, so the order of the additions changes nothing.
template <class F>
__device__ __forceinline__ F shfl_down_fp(const F& x, int off) {
    F r;
    _Pragma("unroll") for (int i = 0; i < F::N; i++) r.v[i] = (uint32_t)__shfl_down((int)x.v[i], off, 64);
    return r;
}
template <class F>
__global__ void __launch_bounds__(256) k_spmv_csr_wave(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const F* __restrict__ coeff,
                                                       size_t n_rows, const F* __restrict__ pub, uint32_t n_inputs, int party,
                                                       const F* __restrict__ wit_a, const F* __restrict__ wit_b, F* __restrict__ out_a, F* __restrict__ out_b) {
    const uint32_t lane = threadIdx.x & 63u;
    for (size_t row = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < n_rows; row += ((size_t)gridDim.x * blockDim.x) >> 6) {
        F acc_a = F::zero(), acc_b = F::zero();
        const uint32_t e = row_ptr[row + 1];
        for (uint32_t k = row_ptr[row] + lane; k < e; k += 64) {
            const uint32_t idx = col[k];
            const F c = ld_fp(coeff + k);
            if (idx < n_inputs) {
                F t = c * ld_fp(pub + idx);
                if (party <= 0) acc_a = acc_a + t;
                else if (party == 1) acc_b = acc_b + t;
            } else {
                acc_a = acc_a + c * ld_fp(wit_a + (idx - n_inputs));
                if (wit_b) acc_b = acc_b + c * ld_fp(wit_b + (idx - n_inputs));
            }
        }
        for (int off = 32; off >= 1; off >>= 1) { acc_a = acc_a + shfl_down_fp(acc_a, off); if (out_b) acc_b = acc_b + shfl_down_fp(acc_b, off); }
        if (lane == 0) { st_fp(out_a + row, acc_a); if (out_b) st_fp(out_b + row, acc_b); }
    }
}

// dst[i] = src[i] for i < n else 0, for dst of length m (building the zero-padded evaluation vectors, groth16.rs:156-171)
template <class F>
__global__ void __launch_bounds__(256) k_zero_tail(F* __restrict__ v, size_t from, size_t to) {
    for (size_t i = from + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < to; i += (size_t)gridDim.x * blockDim.x) st_fp(v + i, F::zero());
}

// ---- co-plonk pointwise helpers (round 2: grand-product polynomial z, co-plonk/src/round2.rs:146-268) ------------------------------
// out[i] = c * a[i] + d   (mul_with_public / add_with_public of a single-component share vector: plain.rs, shamir.rs:471-506)
template <class F>
__global__ void __launch_bounds__(256) k_vec_affine(F* __restrict__ out, const F* __restrict__ a, size_t n, F c, F d) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_fp(out + i, c * ld_fp(a + i) + d);
}
// v[i] = value
template <class F>
__global__ void __launch_bounds__(256) k_vec_fill(F* __restrict__ v, size_t n, F value) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_fp(v + i, value);
}
// out[i] = in[offset + i * stride]   (e.g. every 4th evaluation of a sigma polynomial, round2.rs:196-206)
template <class F>
__global__ void __launch_bounds__(256) k_vec_gather_strided(F* __restrict__ out, const F* __restrict__ in, size_t n, size_t offset, size_t stride) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_fp(out + i, ld_fp(in + offset + i * stride));
}

// out[out_off + i * out_stride] = sum_j coeff[j] * src[j][off[j] + i * stride[j]]   (strides in elements, may be negative)
// The Shamir share algebra as one launch per output: share evaluation (shamir_core.rs:8-31), the Vandermonde step of the
// double-sharing generation (shamir.rs:904-921), the king's interpolation (shamir.rs:330-345), openings (shamir.rs:581-601)
// and reading the LIFO pair buffer backwards (shamir.rs:1012-1025).
template <class F>
__global__ void __launch_bounds__(256) k_vec_lincomb(F* __restrict__ out, long long out_off, long long out_stride, size_t n, LincombArgs<F> a) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        F acc = F::zero();
        for (int j = 0; j < a.n_terms; j++) {
            const F v = ld_fp(a.src[j] + (a.off[j] + (long long)i * a.stride[j]));
            acc = acc + (a.unit[j] ? v : a.coeff[j] * v);
        }
        st_fp(out + (out_off + (long long)i * out_stride), acc);
    }
}

// out[j] = in[idx[j] - base]   (scalars of the non-infinity bases of a compacted table, see cg_bases::compact)
template <class F>
__global__ void __launch_bounds__(256) k_vec_gather_idx(F* __restrict__ out, const F* __restrict__ in, const uint32_t* __restrict__ idx, size_t n, uint32_t base) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_fp(out + i, ld_fp(in + (idx[i] - base)));
}

// Inclusive prefix scan out[i] = in[0] (op) ... (op) in[i], op = field product (what `array_prod_mul` yields for a single-component
// driver, round2.rs:18-41) or field sum (polynomial evaluation and the synthetic division of round 5).  Three launches: tiles of
// 256 x SCAN_ITEMS elements (serial per lane, Hillis-Steele across the workgroup in LDS), a scan of the tile totals by one
// workgroup, and the fix-up pass.
constexpr int SCAN_ITEMS = 8;
template <class F, int OP> struct ScanOp {
    __device__ __forceinline__ static F id() { return OP == 0 ? F::one() : F::zero(); }
    __device__ __forceinline__ static F ap(const F& a, const F& b) { return OP == 0 ? a * b : a + b; }
};
template <class F, int OP>
__device__ __forceinline__ void block_scan(F v, F* sh, F* total) {      // inclusive scan over the 256 lanes, left in sh[]
    const int t = threadIdx.x;
    sh[t] = v; __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        F x = sh[t];
        if (t >= off) x = ScanOp<F, OP>::ap(sh[t - off], x);
        __syncthreads();
        sh[t] = x; __syncthreads();
    }
    if (total) *total = sh[255];
}
template <class F, int OP>
__global__ void __launch_bounds__(256) k_prefix_tiles(F* __restrict__ out, const F* __restrict__ in, size_t n, F* __restrict__ tile_tot) {
    __shared__ F sh[256];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * SCAN_ITEMS;
    F loc[SCAN_ITEMS]; F run = ScanOp<F, OP>::id();
    _Pragma("unroll") for (int k = 0; k < SCAN_ITEMS; k++) { if (base + k < n) run = ScanOp<F, OP>::ap(run, ld_fp(in + base + k)); loc[k] = run; }
    F tot;
    block_scan<F, OP>(run, sh, &tot);
    __syncthreads();
    const F excl = threadIdx.x ? sh[threadIdx.x - 1] : ScanOp<F, OP>::id();
    _Pragma("unroll") for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) st_fp(out + base + k, ScanOp<F, OP>::ap(excl, loc[k]));
    if (threadIdx.x == 0) st_fp(tile_tot + blockIdx.x, tot);
}
template <class F, int OP>
__global__ void __launch_bounds__(256) k_prefix_totals(F* __restrict__ tile_tot, size_t ntiles) {   // in place -> EXCLUSIVE prefix of the tile totals
    __shared__ F sh[256];
    F carry = ScanOp<F, OP>::id();
    for (size_t b0 = 0; b0 < ntiles; b0 += 256) {
        const size_t i = b0 + threadIdx.x;
        F v = i < ntiles ? ld_fp(tile_tot + i) : ScanOp<F, OP>::id();
        F tot;
        block_scan<F, OP>(v, sh, &tot);
        __syncthreads();
        const F excl = ScanOp<F, OP>::ap(carry, threadIdx.x ? sh[threadIdx.x - 1] : ScanOp<F, OP>::id());
        if (i < ntiles) st_fp(tile_tot + i, excl);
        carry = ScanOp<F, OP>::ap(carry, tot);
        __syncthreads();
    }
}
template <class F, int OP>
__global__ void __launch_bounds__(256) k_prefix_fixup(F* __restrict__ out, size_t n, const F* __restrict__ tile_excl) {
    const size_t tile = (size_t)256 * SCAN_ITEMS;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (i < tile) continue;
        st_fp(out + i, ScanOp<F, OP>::ap(ld_fp(tile_excl + i / tile), ld_fp(out + i)));
    }
}
// out[i] = in[i]^-1 (0 -> 0): Montgomery's trick over INV_ITEMS elements per lane, one Fermat inversion per lane (inv_many, plain.rs)
constexpr int INV_ITEMS = 8;
template <class F>
__global__ void __launch_bounds__(128) k_vec_inverse(F* __restrict__ out, const F* __restrict__ in, size_t n) {
    const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * INV_ITEMS;
    if (base >= n) return;
    F x[INV_ITEMS], pre[INV_ITEMS]; F run = F::one();
    _Pragma("unroll") for (int k = 0; k < INV_ITEMS; k++) {
        x[k] = base + k < n ? ld_fp(in + base + k) : F::one();
        pre[k] = run;
        if (!x[k].is_zero()) run = run * x[k];
    }
    F inv = fp_inverse(run);
    _Pragma("unroll") for (int k = INV_ITEMS - 1; k >= 0; k--) {
        if (base + k >= n) continue;
        if (x[k].is_zero()) { st_fp(out + base + k, F::zero()); continue; }
        st_fp(out + base + k, inv * pre[k]);
        inv = inv * x[k];
    }
}

}  // namespace cg
