//! Raw bindings of `include/cogroth16_hip.h` (the subset the drivers use).  Every function returns 0 on success; the message of a
//! failure is `cg_last_error()` (thread local).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

#[repr(C)]
pub struct cg_ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct cg_bases {
    _p: [u8; 0],
}
pub const CG_BN254: i32 = 0;
pub const CG_BLS12_381: i32 = 1;
pub const CG_G1: i32 = 0;
pub const CG_G2: i32 = 1;

extern "C" {
    pub fn cg_last_error() -> *const c_char;
    pub fn cg_ctx_create(device: i32, out: *mut *mut cg_ctx) -> i32;
    pub fn cg_ctx_destroy(ctx: *mut cg_ctx) -> i32;
    pub fn cg_ctx_sync(ctx: *mut cg_ctx) -> i32;
    /// per-context tuning table (include/cogroth16_hip.h: CG_OPT_MSM_CHUNK = 1 .. CG_OPT_MSM_G2_AFTER = 8); never changes results
    pub fn cg_ctx_set_option(ctx: *mut cg_ctx, option: i32, value: i64) -> i32;
    pub fn cg_ctx_get_option(ctx: *const cg_ctx, option: i32, value: *mut i64) -> i32;
    pub fn cg_bases_register(ctx: *mut cg_ctx, curve: i32, group: i32, h_points: *const c_void, n: usize, stride_bytes: usize,
                             infinity_offset: i64, out: *mut *mut cg_bases) -> i32;
    pub fn cg_bases_release(b: *mut cg_bases) -> i32;
    pub fn cg_bases_precompute(ctx: *mut cg_ctx, b: *mut cg_bases, c: i32) -> i32;
    pub fn cg_bases_check_on_curve(ctx: *mut cg_ctx, b: *const cg_bases, n_bad: *mut u64, first_bad: *mut u64) -> i32;
    pub fn cg_bases_check_subgroup(ctx: *mut cg_ctx, b: *const cg_bases, n_bad: *mut u64, first_bad: *mut u64) -> i32;
    /// k scalar vectors (host, Montgomery limbs) against `n` points starting at `offset` of a registered table;
    /// out = k Jacobian points (X, Y, Z), the layout of `ark_ec::short_weierstrass::Projective`
    pub fn cg_msm(ctx: *mut cg_ctx, bases: *const cg_bases, offset: usize, n: usize, h_scalars: *const *const c_void, k: i32,
                  h_out_jacobian: *mut c_void) -> i32;
    /// k host vectors of n elements, in place; `h_group_gen` = the domain's generator as the caller set it (groth16.rs:63-70);
    /// inverse != 0: inverse transform incl. 1/n; `h_coset_gen` (or null) fuses `distribute_powers_and_mul_by_const(v, g, 1)`
    pub fn cg_ntt(ctx: *mut cg_ctx, curve: i32, h_vecs: *const *mut c_void, k: i32, n: usize, h_group_gen: *const c_void, inverse: i32,
                  h_coset_gen: *const c_void) -> i32;
    pub fn cg_vec_mul(ctx: *mut cg_ctx, curve: i32, h_out: *mut c_void, h_a: *const c_void, h_b: *const c_void, n: usize) -> i32;
    /// out = aa*ba + aa*bb + ab*ba + mask (rep3.rs:656-660); mask may be null
    pub fn cg_vec_rep3_mul_local(ctx: *mut cg_ctx, curve: i32, h_out: *mut c_void, h_aa: *const c_void, h_ab: *const c_void,
                                 h_ba: *const c_void, h_bb: *const c_void, h_mask: *const c_void, n: usize) -> i32;
    // device-resident variants (vectors stay on the GPU between trait calls; see DeviceVec in gpu.rs)
    pub fn cg_dev_alloc(ctx: *mut cg_ctx, bytes: usize, d_ptr: *mut *mut c_void) -> i32;
    pub fn cg_dev_free(ctx: *mut cg_ctx, d_ptr: *mut c_void) -> i32;
    pub fn cg_dev_upload(ctx: *mut cg_ctx, d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> i32;
    pub fn cg_dev_download(ctx: *mut cg_ctx, h_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> i32;
    pub fn cg_vec_sub_dev(ctx: *mut cg_ctx, curve: i32, d_out: *mut c_void, d_a: *const c_void, d_b: *const c_void, n: usize) -> i32;
    pub fn cg_vec_add_dev(ctx: *mut cg_ctx, curve: i32, d_out: *mut c_void, d_a: *const c_void, d_b: *const c_void, n: usize) -> i32;
    pub fn cg_vec_distribute_powers_dev(ctx: *mut cg_ctx, curve: i32, d_v: *mut c_void, n: usize, h_g: *const c_void, h_c: *const c_void) -> i32;
    pub fn cg_spmv_csr_dev(ctx: *mut cg_ctx, curve: i32, d_row_ptr: *const u32, d_col: *const u32, d_coeff: *const c_void, n_rows: usize,
                           d_pub: *const c_void, n_pub: u32, party_id: i32, d_wit_a: *const c_void, d_wit_b: *const c_void,
                           d_out_a: *mut c_void, d_out_b: *mut c_void) -> i32;
}

/// message of the last failing call on this thread
pub fn last_error() -> String {
    unsafe {
        let p = cg_last_error();
        if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}
