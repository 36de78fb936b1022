//! The GPU side shared by the three drivers: one `cg_ctx` per party thread (mirrors `&mut self` of the trait methods), the table registry
//! that maps sub-slices of EXPLICITLY registered zkey vectors onto device-resident tables (anything else is uploaded per call), and the layout checks
//! that make it sound to hand arkworks' in-memory values to the C ABI without conversion.
use crate::ffi::*;
use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ff::{Field, PrimeField};
use std::{collections::HashMap, marker::PhantomData, mem::{offset_of, size_of}, os::raw::c_void, ptr};

/// Compile-time (post-monomorphisation) proof that `Affine<P>` / `Projective<P>` have the layout the ABI reads and writes:
/// x at offset 0, y right behind it (the packed `x || y` the kernels load), the infinity flag somewhere after (its offset and the
/// stride are passed to `cg_bases_register`), and (X, Y, Z) contiguous for the Jacobian results.  `repr(Rust)` gives no ordering
/// guarantee, so a toolchain that reorders these fields fails to BUILD instead of proving garbage.
pub struct Layout<P: SWCurveConfig>(PhantomData<P>);
impl<P: SWCurveConfig> Layout<P> {
    pub const COORD: usize = size_of::<P::BaseField>();
    pub const STRIDE: usize = size_of::<Affine<P>>();
    pub const INFINITY_OFFSET: usize = offset_of!(Affine<P>, infinity);
    pub const CHECK: () = {
        assert!(offset_of!(Affine<P>, x) == 0, "Affine.x is not the first field");
        assert!(offset_of!(Affine<P>, y) == size_of::<P::BaseField>(), "Affine.y does not follow x");
        assert!(offset_of!(Affine<P>, infinity) >= 2 * size_of::<P::BaseField>(), "Affine.infinity overlaps the coordinates");
        assert!(offset_of!(Projective<P>, x) == 0 && offset_of!(Projective<P>, y) == size_of::<P::BaseField>()
                    && offset_of!(Projective<P>, z) == 2 * size_of::<P::BaseField>(), "Projective is not (X, Y, Z)");
        assert!(size_of::<Projective<P>>() == 3 * size_of::<P::BaseField>(), "Projective has padding");
        // Fp<MontBackend<_, N>, N> is #[repr(transparent)]-like over BigInt<N>([u64; N]): N limbs, nothing else
        assert!(size_of::<P::ScalarField>() % 8 == 0 && size_of::<P::BaseField>() % 8 == 0);
    };
}

/// BN254 or BLS12-381, decided by the scalar modulus (the ABI's `curve` argument)
pub fn curve_id<F: PrimeField>() -> i32 {
    let m = F::MODULUS;
    if m.as_ref() == <ark_bn254::Fr as PrimeField>::MODULUS.as_ref() {
        CG_BN254
    } else if m.as_ref() == <ark_bls12_381::Fr as PrimeField>::MODULUS.as_ref() {
        CG_BLS12_381
    } else {
        panic!("cogroth16_hip supports the BN254 and BLS12-381 scalar fields only")
    }
}
/// G1 (coordinates in the prime field) or G2 (coordinates in its quadratic extension)
pub fn group_id<P: SWCurveConfig>() -> i32 {
    if <P::BaseField as Field>::extension_degree() == 1 { CG_G1 } else { CG_G2 }
}

pub(crate) fn check(rc: i32, what: &str) {
    if rc != 0 {
        // the trait methods that reach the GPU are infallible in the reference (traits.rs:535-568)
        panic!("cogroth16_hip: {what}: {}", last_error());
    }
}

struct Table {
    bases: *mut cg_bases,
    /// address range of the host slice this table was registered from
    host_lo: usize,
    host_hi: usize,
    stride: usize,
}

pub struct Gpu {
    pub(crate) ctx: *mut cg_ctx,
    tables: Vec<Table>,
    by_start: HashMap<usize, usize>,
    /// window of `cg_bases_precompute` for tables registered from here on (0 = none, -1 = by table size)
    pub precompute: i32,
    /// constraint matrices kept on the device (`Gpu::matrix`)
    matrices: Vec<DeviceMatrix>,
}
// a context is used by one thread at a time; moving the driver to another thread between calls is fine (no thread-local HIP state)
unsafe impl Send for Gpu {}

impl Gpu {
    pub fn new(device: i32) -> eyre::Result<Self> {
        let mut ctx = ptr::null_mut();
        let rc = unsafe { cg_ctx_create(device, &mut ctx) };
        if rc != 0 {
            eyre::bail!("cg_ctx_create({device}): {}", last_error());
        }
        Ok(Self { ctx, tables: Vec::new(), by_start: HashMap::new(), precompute: 0, matrices: Vec::new() })
    }

    /// Registers a whole zkey query (`ZKey::a_query` … `h_query`, circom-types/src/groth16/zkey.rs:48-71) once.  The vectors live for the
    /// process and are reused across proofs; later `msm_public_points` calls on any sub-slice of them resolve to (table, offset).
    pub fn register<P: SWCurveConfig>(&mut self, points: &[Affine<P>]) -> usize {
        #[allow(clippy::let_unit_value)]
        let _ = Layout::<P>::CHECK;
        let lo = points.as_ptr() as usize;
        if let Some(&i) = self.by_start.get(&lo) {
            // the same vector registered twice (same start, same length): the caller's promise about its lifetime covers both
            if self.tables[i].host_hi == lo + points.len() * Layout::<P>::STRIDE {
                return i;
            }
        }
        let mut bases = ptr::null_mut();
        check(
            unsafe {
                cg_bases_register(self.ctx, curve_id::<P::ScalarField>(), group_id::<P>(), points.as_ptr() as *const c_void, points.len(),
                                  Layout::<P>::STRIDE, Layout::<P>::INFINITY_OFFSET as i64, &mut bases)
            },
            "cg_bases_register",
        );
        if self.precompute != 0 && points.len() >= (1 << 14) {
            check(unsafe { cg_bases_precompute(self.ctx, bases, if self.precompute < 0 { 0 } else { self.precompute }) }, "cg_bases_precompute");
        }
        self.tables.push(Table { bases, host_lo: lo, host_hi: lo + points.len() * Layout::<P>::STRIDE, stride: Layout::<P>::STRIDE });
        self.by_start.insert(lo, self.tables.len() - 1);
        self.tables.len() - 1
    }

    /// The parser's per-point checks (circom-types/src/traits.rs:107-155) on the device, for zkeys parsed with `Validate::No`
    pub fn validate(&mut self, table: usize) -> Result<(), String> {
        let (mut bad, mut first) = (0u64, 0u64);
        check(unsafe { cg_bases_check_on_curve(self.ctx, self.tables[table].bases, &mut bad, &mut first) }, "cg_bases_check_on_curve");
        if bad != 0 {
            return Err(format!("point {first} is not on the curve ({bad} bad points)"));
        }
        check(unsafe { cg_bases_check_subgroup(self.ctx, self.tables[table].bases, &mut bad, &mut first) }, "cg_bases_check_subgroup");
        if bad != 0 {
            return Err(format!("point {first} is not in the correct subgroup ({bad} bad points)"));
        }
        Ok(())
    }

    /// Forgets a table registered with `register` (its host vector is about to be dropped or rewritten) and frees its device copy.
    /// Indices of other tables stay valid.
    pub fn unregister(&mut self, table: usize) {
        let t = &mut self.tables[table];
        if !t.bases.is_null() {
            unsafe { cg_bases_release(t.bases) };
            t.bases = ptr::null_mut();
            self.by_start.remove(&t.host_lo);
            t.host_lo = 0;
            t.host_hi = 0;
        }
    }

    /// (table, offset in points) for a slice that lies inside a vector registered EXPLICITLY with `register` — the provers pass
    /// sub-slices such as `&query[1 + pub_len..]` (groth16.rs:221) and `&p_tau[..len]` (co-plonk round1.rs:276-290).  Only explicit
    /// registrations are matched: the caller vouches that those vectors live, unchanged, as long as the table does (zkey vectors do,
    /// zkey.rs:48-71).  An address range says nothing about the contents of any other slice, so nothing else is ever cached.
    fn registered<P: SWCurveConfig>(&self, points: &[Affine<P>]) -> Option<(*const cg_bases, usize)> {
        let lo = points.as_ptr() as usize;
        let hi = lo + points.len() * Layout::<P>::STRIDE;
        self.tables
            .iter()
            .find(|t| !t.bases.is_null() && t.host_lo <= lo && hi <= t.host_hi && t.stride == Layout::<P>::STRIDE && (lo - t.host_lo) % t.stride == 0)
            .map(|t| (t.bases as *const cg_bases, (lo - t.host_lo) / t.stride))
    }

    /// `k` MSMs over the same points (one per share component, the loop of rep3.rs:942-943 in one call: the points are gathered once
    /// per component from the same resident table and the components' digit schedules overlap with each other's accumulation).
    /// Points that are not part of a registered vector are uploaded as a temporary table that is released before the call returns.
    pub fn msm<P: SWCurveConfig>(&mut self, points: &[Affine<P>], scalars: &[&[P::ScalarField]]) -> Vec<Projective<P>> {
        #[allow(clippy::let_unit_value)]
        let _ = Layout::<P>::CHECK;
        for s in scalars {
            assert_eq!(s.len(), points.len(), "msm_public_points: length mismatch");
        }
        let mut temporary = ptr::null_mut();
        let (bases, offset) = match self.registered(points) {
            Some(hit) => hit,
            None => {
                check(
                    unsafe {
                        cg_bases_register(self.ctx, curve_id::<P::ScalarField>(), group_id::<P>(), points.as_ptr() as *const c_void, points.len(),
                                          Layout::<P>::STRIDE, Layout::<P>::INFINITY_OFFSET as i64, &mut temporary)
                    },
                    "cg_bases_register",
                );
                (temporary as *const cg_bases, 0)
            }
        };
        let ptrs: Vec<*const c_void> = scalars.iter().map(|s| s.as_ptr() as *const c_void).collect();
        let mut out = vec![Projective::<P>::default(); scalars.len()];
        tracing::trace!("> MSM public points for {} elements", points.len()); // the reference's enter / exit events (rep3.rs:940-944)
        let rc = unsafe { cg_msm(self.ctx, bases, offset, points.len(), ptrs.as_ptr(), ptrs.len() as i32, out.as_mut_ptr() as *mut c_void) };
        if !temporary.is_null() {
            unsafe { cg_bases_release(temporary) };
        }
        check(rc, "cg_msm");
        tracing::trace!("< MSM public points for {} elements", points.len());
        out
    }

    /// in-place transforms of `k` vectors with the domain's own generator (the callers overwrite `group_gen`, groth16.rs:63-70,
    /// co-plonk/src/types.rs:83-90, so nothing about the root of unity is assumed here)
    pub fn ntt<F: PrimeField>(&mut self, vecs: &mut [&mut [F]], group_gen: F, inverse: bool, coset_gen: Option<F>) {
        let n = vecs[0].len();
        assert!(vecs.iter().all(|v| v.len() == n) && n.is_power_of_two(), "fft: vectors must have the domain's (power of two) size");
        let ptrs: Vec<*mut c_void> = vecs.iter_mut().map(|v| v.as_mut_ptr() as *mut c_void).collect();
        let g = coset_gen.as_ref().map_or(ptr::null(), |g| g as *const F as *const c_void);
        let what = if inverse { "IFFT" } else { "FFT" }; // the reference's enter / exit events (rep3.rs:886-897,905-920)
        tracing::trace!("> {what} (in place) for {n} elements");
        check(
            unsafe { cg_ntt(self.ctx, curve_id::<F>(), ptrs.as_ptr(), ptrs.len() as i32, n, &group_gen as *const F as *const c_void, inverse as i32, g) },
            "cg_ntt",
        );
        tracing::trace!("< {what} (in place) for {n} elements");
    }

    pub fn mul<F: PrimeField>(&mut self, a: &[F], b: &[F]) -> Vec<F> {
        assert_eq!(a.len(), b.len());
        let mut out = vec![F::zero(); a.len()];
        check(
            unsafe { cg_vec_mul(self.ctx, curve_id::<F>(), out.as_mut_ptr() as *mut c_void, a.as_ptr() as *const c_void, b.as_ptr() as *const c_void, a.len()) },
            "cg_vec_mul",
        );
        out
    }

    /// aa*ba + aa*bb + ab*ba + mask, the local part of REP3 `mul_vec` (rep3.rs:656-660)
    pub fn rep3_mul_local<F: PrimeField>(&mut self, aa: &[F], ab: &[F], ba: &[F], bb: &[F], mask: &[F]) -> Vec<F> {
        let n = aa.len();
        assert!(ab.len() == n && ba.len() == n && bb.len() == n && mask.len() == n);
        let mut out = vec![F::zero(); n];
        check(
            unsafe {
                cg_vec_rep3_mul_local(self.ctx, curve_id::<F>(), out.as_mut_ptr() as *mut c_void, aa.as_ptr() as *const c_void,
                                      ab.as_ptr() as *const c_void, ba.as_ptr() as *const c_void, bb.as_ptr() as *const c_void,
                                      mask.as_ptr() as *const c_void, n)
            },
            "cg_vec_rep3_mul_local",
        );
        out
    }
}

// ---- device-resident helpers of the trait-level drivers (round 6: the per-call drivers are not host-bound any more) ---------------------------
/// A device block that is given back (parked for reuse, cg_dev_free) when it goes out of scope.
struct DevBuf {
    ctx: *mut cg_ctx,
    p: *mut c_void,
}
impl DevBuf {
    fn new(ctx: *mut cg_ctx, bytes: usize) -> Self {
        let mut p = ptr::null_mut();
        check(unsafe { cg_dev_alloc(ctx, bytes.max(32), &mut p) }, "cg_dev_alloc");
        Self { ctx, p }
    }
    fn from_slice<T>(ctx: *mut cg_ctx, v: &[T]) -> Self {
        let b = Self::new(ctx, std::mem::size_of_val(v));
        if !v.is_empty() {
            check(unsafe { cg_dev_upload(ctx, b.p, v.as_ptr() as *const c_void, std::mem::size_of_val(v)) }, "cg_dev_upload");
        }
        b
    }
}
impl Drop for DevBuf {
    fn drop(&mut self) {
        unsafe { cg_dev_free(self.ctx, self.p) };
    }
}

/// A constraint matrix resident on the device in CSR form (`ConstraintMatrices::a` / `::b` are `Vec<Vec<(F, usize)>>`, one inner vector per
/// constraint): flattened and uploaded ONCE per zkey, then every proof evaluates all of its rows in one `cg_spmv_csr_dev` launch
/// instead of `num_constraints` calls of `evaluate_constraint` on the host (groth16.rs:159-166).
pub struct DeviceMatrix {
    row_ptr: DevBuf,
    col: DevBuf,
    coeff: DevBuf,
    rows: usize,
    /// address and length of the host matrix this copy was made from (the cache key of `Gpu::matrix`)
    host: (usize, usize),
}

impl Gpu {
    /// The resident copy of `rows` (made on first use; the zkey's matrices live, unchanged, as long as the zkey does).
    pub fn matrix<F: PrimeField>(&mut self, rows: &[Vec<(F, usize)>]) -> usize {
        let key = (rows.as_ptr() as usize, rows.len());
        if let Some(i) = self.matrices.iter().position(|m| m.host == key) {
            return i;
        }
        let nnz: usize = rows.iter().map(Vec::len).sum();
        assert!(nnz < u32::MAX as usize, "constraint matrix with 2^32 or more entries");
        let (mut row_ptr, mut col, mut coeff) = (Vec::with_capacity(rows.len() + 1), Vec::with_capacity(nnz), Vec::with_capacity(nnz));
        row_ptr.push(0u32);
        for r in rows {
            for (c, i) in r {
                coeff.push(*c);
                col.push(u32::try_from(*i).expect("signal index above 2^32"));
            }
            row_ptr.push(col.len() as u32);
        }
        let m = DeviceMatrix { row_ptr: DevBuf::from_slice(self.ctx, &row_ptr), col: DevBuf::from_slice(self.ctx, &col), coeff: DevBuf::from_slice(self.ctx, &coeff), rows: rows.len(), host: key };
        self.matrices.push(m);
        self.matrices.len() - 1
    }

    /// `evaluate_constraint` for EVERY row of a resident matrix (traits.rs:180 / rep3.rs:690-708 with the add_with_public asymmetry of
    /// rep3.rs:600-608): `party` = -1 for one share component (plain, Shamir: `wit_b` empty), 0..2 = the REP3 party id.  Returns the two
    /// component vectors, `domain_size` long (rows past the matrix stay zero, as `vec![FieldShare::default(); domain_size]` leaves them).
    pub fn evaluate_constraints<F: PrimeField>(&mut self, matrix: usize, domain_size: usize, public_inputs: &[F], party: i32, wit_a: &[F], wit_b: &[F]) -> (Vec<F>, Vec<F>) {
        let m = &self.matrices[matrix];
        assert!(m.rows <= domain_size && (party < 0 || wit_b.len() == wit_a.len()));
        let bytes = domain_size * size_of::<F>();
        let (d_pub, d_a, d_b) = (DevBuf::from_slice(self.ctx, public_inputs), DevBuf::from_slice(self.ctx, wit_a), DevBuf::from_slice(self.ctx, wit_b));
        let (o_a, o_b) = (DevBuf::new(self.ctx, bytes), DevBuf::new(self.ctx, bytes));
        let two = party >= 0;
        check(unsafe { cg_dev_memset_zero(self.ctx, o_a.p, bytes) }, "cg_dev_memset_zero");
        if two {
            check(unsafe { cg_dev_memset_zero(self.ctx, o_b.p, bytes) }, "cg_dev_memset_zero");
        }
        check(
            unsafe {
                cg_spmv_csr_dev(self.ctx, curve_id::<F>(), m.row_ptr.p as *const u32, m.col.p as *const u32, m.coeff.p, m.rows, d_pub.p, public_inputs.len() as u32, party,
                                d_a.p, if two { d_b.p as *const c_void } else { ptr::null() }, o_a.p, if two { o_b.p } else { ptr::null_mut() })
            },
            "cg_spmv_csr_dev",
        );
        let mut out_a = vec![F::zero(); domain_size];
        let mut out_b = vec![F::zero(); if two { domain_size } else { 0 }];
        check(unsafe { cg_dev_download(self.ctx, out_a.as_mut_ptr() as *mut c_void, o_a.p, bytes) }, "cg_dev_download");
        if two {
            check(unsafe { cg_dev_download(self.ctx, out_b.as_mut_ptr() as *mut c_void, o_b.p, bytes) }, "cg_dev_download");
        }
        (out_a, out_b)
    }

    /// The local part of REP3 `mul_vec` (rep3.rs:656-660) with the masks DRAWN ON THE DEVICE: n x `F::rand` from each of the party's two
    /// ChaCha12 generators, described by seed and 32-bit word position (`ChaCha12Rng::get_seed` / `get_word_pos`), mask_i = r1_i - r2_i
    /// (rngs.rs:37-40) — the draws the stock implementation makes one by one on a host thread (~70 ns each: 0.6 s per proof at 2^22).
    /// Returns the masked local products and the positions both generators have to be set to (`set_word_pos`) so that the next host draw
    /// is the one the stock implementation would make.
    #[allow(clippy::too_many_arguments)]
    pub fn rep3_mul_local_drawn<F: PrimeField>(&mut self, aa: &[F], ab: &[F], ba: &[F], bb: &[F], seed1: &[u8; 32], pos1: u64, seed2: &[u8; 32], pos2: u64) -> (Vec<F>, u64, u64) {
        let n = aa.len();
        assert!(ab.len() == n && ba.len() == n && bb.len() == n);
        let bytes = n * size_of::<F>();
        let curve = curve_id::<F>();
        let (d_aa, d_ab, d_ba, d_bb) = (DevBuf::from_slice(self.ctx, aa), DevBuf::from_slice(self.ctx, ab), DevBuf::from_slice(self.ctx, ba), DevBuf::from_slice(self.ctx, bb));
        let (r1, r2) = (DevBuf::new(self.ctx, bytes), DevBuf::new(self.ctx, bytes));
        let (mut after1, mut after2) = (0u64, 0u64);
        check(unsafe { cg_chacha12_fr_rand_dev(self.ctx, curve, seed1.as_ptr(), pos1, n, r1.p, &mut after1) }, "cg_chacha12_fr_rand_dev");
        check(unsafe { cg_chacha12_fr_rand_dev(self.ctx, curve, seed2.as_ptr(), pos2, n, r2.p, &mut after2) }, "cg_chacha12_fr_rand_dev");
        check(unsafe { cg_vec_sub_dev(self.ctx, curve, r1.p, r1.p, r2.p, n) }, "cg_vec_sub_dev");
        check(unsafe { cg_vec_rep3_mul_local_dev(self.ctx, curve, r2.p, d_aa.p, d_ab.p, d_ba.p, d_bb.p, r1.p, n) }, "cg_vec_rep3_mul_local_dev");
        let mut out = vec![F::zero(); n];
        check(unsafe { cg_dev_download(self.ctx, out.as_mut_ptr() as *mut c_void, r2.p, bytes) }, "cg_dev_download");
        (out, after1, after2)
    }
}

impl Drop for Gpu {
    fn drop(&mut self) {
        self.matrices.clear(); // (their blocks go back through the context that is destroyed below)
        unsafe {
            for t in &self.tables {
                if !t.bases.is_null() {
                    cg_bases_release(t.bases);
                }
            }
            cg_ctx_destroy(self.ctx);
        }
    }
}
