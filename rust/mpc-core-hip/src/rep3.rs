//! `Rep3HipProtocol`: the REP3 driver with the O(n) work on the GPU.  Network rounds, the correlated randomness (ChaCha12 streams whose
//! draws must stay byte-compatible with the peers', rep3/rngs.rs:37-46) and every O(1) operation are the stock `Rep3Protocol`'s.
//! Needs the two accessors of `rust/mpc-core-accessors.patch` (`network_mut`, `masking_field_elements`): the fields are private.
use crate::gpu::Gpu;
use ark_ec::{pairing::Pairing, short_weierstrass::{Projective, SWCurveConfig}, CurveGroup};
use ark_ff::PrimeField;
use ark_poly::EvaluationDomain;
use mpc_core::{
    protocols::rep3::{fieldshare::Rep3PrimeFieldShareVec, network::Rep3Network, pointshare::Rep3PointShare, Rep3PrimeFieldShare, Rep3Protocol},
    traits::{EcMpcProtocol, FFTProvider, FieldShareVecTrait, MSMProvider, PairingEcMpcProtocol, PrimeFieldMpcProtocol},
};
use std::io::Result as IoResult;

/// `mul_vec` calls of at least this many elements draw their masks on the GPU (a host draw is ~70 ns per element, a device call ~60 us:
/// the C++ mirror's threshold, host/driver.hpp DEVICE_MASKS_MIN)
pub const DEVICE_MASKS_MIN: usize = 1 << 11;

pub struct Rep3HipProtocol<F: PrimeField, N: Rep3Network> {
    inner: Rep3Protocol<F, N>,
    gpu: Gpu,
}

impl<F: PrimeField, N: Rep3Network> Rep3HipProtocol<F, N> {
    /// `Rep3Protocol::new` (rep3.rs:385-398: seeds the correlated randomness over the network) + a GPU context
    pub fn new(network: N, device: i32) -> eyre::Result<Self> {
        Ok(Self { inner: Rep3Protocol::new(network)?, gpu: Gpu::new(device)? })
    }
    /// the table cache (register the zkey's queries once, right after parsing: co-circom.rs:482)
    pub fn gpu(&mut self) -> &mut Gpu {
        &mut self.gpu
    }
    pub fn into_inner(self) -> Rep3Protocol<F, N> {
        self.inner
    }
}

// ---- PrimeFieldMpcProtocol (traits.rs:43-223): vector methods on the GPU, the rest forwarded -------------------------------------------
impl<F: PrimeField, N: Rep3Network> PrimeFieldMpcProtocol<F> for Rep3HipProtocol<F, N> {
    type FieldShare = Rep3PrimeFieldShare<F>;
    type FieldShareVec = Rep3PrimeFieldShareVec<F>;

    fn add(&mut self, a: &Self::FieldShare, b: &Self::FieldShare) -> Self::FieldShare { self.inner.add(a, b) }
    fn sub(&mut self, a: &Self::FieldShare, b: &Self::FieldShare) -> Self::FieldShare { self.inner.sub(a, b) }
    fn add_with_public(&mut self, a: &F, b: &Self::FieldShare) -> Self::FieldShare { self.inner.add_with_public(a, b) }
    /// O(n) but memory bound and two cache-resident passes on the host: it stays there unless the operands are already on the device
    /// (the C++ host mirror keeps them resident and uses cg_vec_sub_dev; with host `Vec`s a PCIe round trip would cost more)
    fn sub_assign_vec(&mut self, a: &mut Self::FieldShareVec, b: &Self::FieldShareVec) { self.inner.sub_assign_vec(a, b) }
    fn mul(&mut self, a: &Self::FieldShare, b: &Self::FieldShare) -> IoResult<Self::FieldShare> { self.inner.mul(a, b) }
    fn mul_many(&mut self, a: &[Self::FieldShare], b: &[Self::FieldShare]) -> IoResult<Vec<Self::FieldShare>> { self.inner.mul_many(a, b) }
    fn mul_with_public(&mut self, a: &F, b: &Self::FieldShare) -> Self::FieldShare { self.inner.mul_with_public(a, b) }
    fn inv(&mut self, a: &Self::FieldShare) -> IoResult<Self::FieldShare> { self.inner.inv(a) }
    fn inv_many(&mut self, a: &[Self::FieldShare]) -> IoResult<Vec<Self::FieldShare>> { self.inner.inv_many(a) }
    fn inv_many_in_place(&mut self, a: &mut [Self::FieldShare]) -> IoResult<()> { self.inner.inv_many_in_place(a) }
    fn neg(&mut self, a: &Self::FieldShare) -> Self::FieldShare { self.inner.neg(a) }
    fn neg_vec_in_place(&mut self, a: &mut Self::FieldShareVec) { self.inner.neg_vec_in_place(a) }
    fn neg_vec_in_place_limit(&mut self, a: &mut Self::FieldShareVec, limit: usize) { self.inner.neg_vec_in_place_limit(a, limit) }
    fn rand(&mut self) -> IoResult<Self::FieldShare> { self.inner.rand() }
    fn open(&mut self, a: &Self::FieldShare) -> IoResult<F> { self.inner.open(a) }
    fn open_many(&mut self, a: &[Self::FieldShare]) -> IoResult<Vec<F>> { self.inner.open_many(a) }
    fn add_vec(&mut self, a: &Self::FieldShareVec, b: &Self::FieldShareVec) -> Self::FieldShareVec { self.inner.add_vec(a, b) }

    /// rep3.rs:650-670.  Same values, same messages.  From `DEVICE_MASKS_MIN` elements on the masks are drawn ON THE GPU from the party's
    /// own two ChaCha12 generators — read by seed and word position, both set behind the draws afterwards (`rand_stream_state` /
    /// `set_rand_stream_positions`, rust/mpc-core-accessors.patch), so the next draw of the stock implementation is the one it would
    /// have made — instead of n `masking_field_element` calls on this thread (~70 ns each: 0.6 s per proof at 2^22, more than the GPU
    /// needs for the whole proof).  Below the threshold the masks are drawn on the host in the reference's order, as before.
    fn mul_vec(&mut self, a: &Self::FieldShareVec, b: &Self::FieldShareVec) -> IoResult<Self::FieldShareVec> {
        debug_assert_eq!(a.get_len(), b.get_len());
        let n = a.get_len();
        let (aa, ab) = a.clone().get_ab();
        let (ba, bb) = b.clone().get_ab();
        let local_a = if n >= DEVICE_MASKS_MIN {
            let (seed1, pos1, seed2, pos2) = self.inner.rand_stream_state();
            // (a ChaCha12Rng stream is 2^68 words long; the device addresses the first 2^64 of them — 2^59 draws)
            let (pos1, pos2) = (u64::try_from(pos1).expect("stream position above 2^64"), u64::try_from(pos2).expect("stream position above 2^64"));
            let (local_a, after1, after2) = self.gpu.rep3_mul_local_drawn(&aa, &ab, &ba, &bb, &seed1, pos1, &seed2, pos2);
            self.inner.set_rand_stream_positions(after1 as u128, after2 as u128);
            local_a
        } else {
            let mask: Vec<F> = self.inner.masking_field_elements(n);
            self.gpu.rep3_mul_local(&aa, &ab, &ba, &bb, &mask)
        };
        self.inner.network_mut().send_next_many(&local_a)?;
        let local_b: Vec<F> = self.inner.network_mut().recv_prev_many()?;
        if local_b.len() != local_a.len() {
            return Err(std::io::Error::new(std::io::ErrorKind::InvalidData, "During execution of mul_vec in MPC: Invalid number of elements received"));
        }
        Ok(Rep3PrimeFieldShareVec::new(local_a, local_b))
    }

    fn promote_to_trivial_share(&self, public_values: F) -> Self::FieldShare { self.inner.promote_to_trivial_share(public_values) }
    fn promote_to_trivial_shares(&self, public_values: &[F]) -> Self::FieldShareVec { self.inner.promote_to_trivial_shares(public_values) }
    /// rep3.rs:681-688 is a serial running product on the CPU; with the transforms on the GPU the provers' sequence
    /// `ifft_in_place; distribute_powers_and_mul_by_const(g, 1); fft_in_place` is better served by `ifft_coset_in_place` below
    /// (one kernel less, no host pass); called on its own it stays on the host.
    fn distribute_powers_and_mul_by_const(&mut self, coeffs: &mut Self::FieldShareVec, g: F, c: F) { self.inner.distribute_powers_and_mul_by_const(coeffs, g, c) }
    /// one row of the constraint matrices, on the host with the reference's semantics.  The prover's loop over all rows
    /// (groth16.rs:159-166) goes through `evaluate_constraints` below instead (rust/co-groth16-evaluate-constraints.patch).
    fn evaluate_constraint(&mut self, lhs: &[(F, usize)], public_inputs: &[F], private_witness: &Self::FieldShareVec) -> Self::FieldShare {
        self.inner.evaluate_constraint(lhs, public_inputs, private_witness)
    }
    /// ALL rows of one constraint matrix in one launch (the matrix is flattened to CSR and kept on the device at its first use):
    /// overrides the default body the patch gives the trait method (the row-by-row loop).
    fn evaluate_constraints(&mut self, matrix: &[Vec<(F, usize)>], domain_size: usize, public_inputs: &[F], private_witness: &Self::FieldShareVec) -> Self::FieldShareVec {
        let (wit_a, wit_b) = private_witness.clone().get_ab();
        let id = self.inner.network_mut().get_id() as usize as i32;      // PartyID::ID0..ID2 (rep3/id.rs): who adds the public inputs (rep3.rs:600-608)
        let m = self.gpu.matrix(matrix);
        let (a, b) = self.gpu.evaluate_constraints(m, domain_size, public_inputs, id, &wit_a, &wit_b);
        Rep3PrimeFieldShareVec::new(a, b)
    }
    fn clone_from_slice(&self, dst: &mut Self::FieldShareVec, src: &Self::FieldShareVec, dst_offset: usize, src_offset: usize, len: usize) {
        self.inner.clone_from_slice(dst, src, dst_offset, src_offset, len)
    }
    fn mul_open(&mut self, a: &Self::FieldShare, b: &Self::FieldShare) -> IoResult<F> { self.inner.mul_open(a, b) }
    fn mul_open_many(&mut self, a: &[Self::FieldShare], b: &[Self::FieldShare]) -> IoResult<Vec<F>> { self.inner.mul_open_many(a, b) }
}

// ---- EcMpcProtocol / PairingEcMpcProtocol (traits.rs:472-533): O(1) point algebra and openings, forwarded --------------------------------
impl<C: CurveGroup, N: Rep3Network> EcMpcProtocol<C> for Rep3HipProtocol<C::ScalarField, N> {
    type PointShare = Rep3PointShare<C>;
    fn add_points(&mut self, a: &Self::PointShare, b: &Self::PointShare) -> Self::PointShare { EcMpcProtocol::<C>::add_points(&mut self.inner, a, b) }
    fn sub_points(&mut self, a: &Self::PointShare, b: &Self::PointShare) -> Self::PointShare { EcMpcProtocol::<C>::sub_points(&mut self.inner, a, b) }
    fn add_assign_points(&mut self, a: &mut Self::PointShare, b: &Self::PointShare) { EcMpcProtocol::<C>::add_assign_points(&mut self.inner, a, b) }
    fn sub_assign_points(&mut self, a: &mut Self::PointShare, b: &Self::PointShare) { EcMpcProtocol::<C>::sub_assign_points(&mut self.inner, a, b) }
    fn add_assign_points_public(&mut self, a: &mut Self::PointShare, b: &C) { EcMpcProtocol::<C>::add_assign_points_public(&mut self.inner, a, b) }
    fn sub_assign_points_public(&mut self, a: &mut Self::PointShare, b: &C) { EcMpcProtocol::<C>::sub_assign_points_public(&mut self.inner, a, b) }
    fn add_assign_points_public_affine(&mut self, a: &mut Self::PointShare, b: &C::Affine) { EcMpcProtocol::<C>::add_assign_points_public_affine(&mut self.inner, a, b) }
    fn sub_assign_points_public_affine(&mut self, a: &mut Self::PointShare, b: &C::Affine) { EcMpcProtocol::<C>::sub_assign_points_public_affine(&mut self.inner, a, b) }
    fn scalar_mul_public_point(&mut self, a: &C, b: &Self::FieldShare) -> Self::PointShare { EcMpcProtocol::<C>::scalar_mul_public_point(&mut self.inner, a, b) }
    fn scalar_mul_public_scalar(&mut self, a: &Self::PointShare, b: &C::ScalarField) -> Self::PointShare { EcMpcProtocol::<C>::scalar_mul_public_scalar(&mut self.inner, a, b) }
    fn scalar_mul(&mut self, a: &Self::PointShare, b: &Self::FieldShare) -> IoResult<Self::PointShare> { EcMpcProtocol::<C>::scalar_mul(&mut self.inner, a, b) }
    fn open_point(&mut self, a: &Self::PointShare) -> IoResult<C> { EcMpcProtocol::<C>::open_point(&mut self.inner, a) }
    fn open_point_many(&mut self, a: &[Self::PointShare]) -> IoResult<Vec<C>> { EcMpcProtocol::<C>::open_point_many(&mut self.inner, a) }
}
impl<P: Pairing, N: Rep3Network> PairingEcMpcProtocol<P> for Rep3HipProtocol<P::ScalarField, N> {
    fn open_two_points(&mut self, a: &<Self as EcMpcProtocol<P::G1>>::PointShare, b: &<Self as EcMpcProtocol<P::G2>>::PointShare) -> IoResult<(P::G1, P::G2)> {
        PairingEcMpcProtocol::<P>::open_two_points(&mut self.inner, a, b)
    }
}

// ---- FFTProvider (traits.rs:535-558 / rep3.rs:887-931): both share components in one call ----------------------------------------------------
impl<F: PrimeField, N: Rep3Network> FFTProvider<F> for Rep3HipProtocol<F, N> {
    fn fft<D: EvaluationDomain<F>>(&mut self, data: Self::FieldShareVec, domain: &D) -> Self::FieldShareVec {
        let mut data = pad(data, domain.size());
        self.fft_in_place(&mut data, domain);
        data
    }
    fn fft_in_place<D: EvaluationDomain<F>>(&mut self, data: &mut Self::FieldShareVec, domain: &D) {
        assert!(domain.coset_offset().is_one(), "cogroth16_hip: transforms over a coset domain are not wired (the provers use offset 1)");
        with_ab(data, domain.size(), |a, b| self.gpu.ntt(&mut [a, b], domain.group_gen(), false, None));
    }
    fn ifft<D: EvaluationDomain<F>>(&mut self, data: &Self::FieldShareVec, domain: &D) -> Self::FieldShareVec {
        let mut data = pad(data.clone(), domain.size());
        self.ifft_in_place(&mut data, domain);
        data
    }
    fn ifft_in_place<D: EvaluationDomain<F>>(&mut self, data: &mut Self::FieldShareVec, domain: &D) {
        assert!(domain.coset_offset().is_one(), "cogroth16_hip: transforms over a coset domain are not wired (the provers use offset 1)");
        with_ab(data, domain.size(), |a, b| self.gpu.ntt(&mut [a, b], domain.group_gen(), true, None));
    }
    /// O(n) Horner evaluation, co-plonk only (rep3.rs:923-931); the host mirror runs it as a prefix product on the device
    fn evaluate_poly_public(&mut self, poly: Self::FieldShareVec, point: &F) -> Self::FieldShare { self.inner.evaluate_poly_public(poly, point) }
}
impl<F: PrimeField, N: Rep3Network> Rep3HipProtocol<F, N> {
    /// `ifft_in_place` + `distribute_powers_and_mul_by_const(g, 1)` in one launch sequence (groth16.rs:175-186 does the two back to back)
    pub fn ifft_coset_in_place<D: EvaluationDomain<F>>(&mut self, data: &mut Rep3PrimeFieldShareVec<F>, domain: &D, g: F) {
        with_ab(data, domain.size(), |a, b| self.gpu.ntt(&mut [a, b], domain.group_gen(), true, Some(g)));
    }
}

// ---- MSMProvider (traits.rs:561-568 / rep3.rs:934-947) ---------------------------------------------------------------------------------------
impl<Q: SWCurveConfig, N: Rep3Network> MSMProvider<Projective<Q>> for Rep3HipProtocol<Q::ScalarField, N> {
    fn msm_public_points(&mut self, points: &[<Projective<Q> as CurveGroup>::Affine], scalars: &Self::FieldShareVec) -> Self::PointShare {
        debug_assert_eq!(points.len(), scalars.get_len());
        let (a, b) = scalars.clone().get_ab();           // get_ab is the only public view of the two component vectors
        let mut r = self.gpu.msm::<Q>(points, &[&a, &b]).into_iter();
        Rep3PointShare::new(r.next().unwrap(), r.next().unwrap())
    }
}

// `Rep3PrimeFieldShareVec` exposes its components only by value (`get_ab`, fieldshare.rs:245-248): take them out, work, put them back
fn with_ab<F: PrimeField>(v: &mut Rep3PrimeFieldShareVec<F>, n: usize, f: impl FnOnce(&mut [F], &mut [F])) {
    let (mut a, mut b) = std::mem::take(v).get_ab();
    a.resize(n, F::zero());
    b.resize(n, F::zero());
    f(&mut a, &mut b);
    *v = Rep3PrimeFieldShareVec::new(a, b);
}
fn pad<F: PrimeField>(v: Rep3PrimeFieldShareVec<F>, n: usize) -> Rep3PrimeFieldShareVec<F> {
    let (mut a, mut b) = v.get_ab();
    a.resize(n, F::zero());
    b.resize(n, F::zero());
    Rep3PrimeFieldShareVec::new(a, b)
}
