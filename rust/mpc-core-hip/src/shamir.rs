//! `ShamirHipProtocol`: transforms and MSMs of the Shamir driver (shamir.rs:826-871, 1027-1039) on the GPU.  The vector multiplication
//! (`mul_vec` = local product + `degree_reduce_vec`, shamir.rs:302-384, 618-621) keeps the stock implementation here: its local
//! part needs the protocol's private double-sharing buffer; the C++ host mirror shows the device version (local product, double
//! sharings generated with `cg_vec_lincomb_dev`, king-side Lagrange combination on the device).
use crate::gpu::Gpu;
use ark_ec::{pairing::Pairing, short_weierstrass::{Projective, SWCurveConfig}, CurveGroup};
use ark_ff::PrimeField;
use ark_poly::EvaluationDomain;
use mpc_core::{
    protocols::shamir::{fieldshare::{ShamirPrimeFieldShare, ShamirPrimeFieldShareVec}, network::ShamirNetwork, pointshare::ShamirPointShare, ShamirProtocol},
    traits::{EcMpcProtocol, FFTProvider, FieldShareVecTrait, MSMProvider, PairingEcMpcProtocol, PrimeFieldMpcProtocol},
};
use std::io::Result as IoResult;

pub struct ShamirHipProtocol<F: PrimeField, N: ShamirNetwork> {
    inner: ShamirProtocol<F, N>,
    gpu: Gpu,
}
impl<F: PrimeField, N: ShamirNetwork> ShamirHipProtocol<F, N> {
    pub fn new(inner: ShamirProtocol<F, N>, device: i32) -> eyre::Result<Self> {
        Ok(Self { inner, gpu: Gpu::new(device)? })
    }
    pub fn gpu(&mut self) -> &mut Gpu {
        &mut self.gpu
    }
}
impl<F: PrimeField, N: ShamirNetwork> PrimeFieldMpcProtocol<F> for ShamirHipProtocol<F, N> {
    type FieldShare = ShamirPrimeFieldShare<F>;
    type FieldShareVec = ShamirPrimeFieldShareVec<F>;
    fn add(&mut self, a: &Self::FieldShare, b: &Self::FieldShare) -> Self::FieldShare { self.inner.add(a, b) }
    fn sub(&mut self, a: &Self::FieldShare, b: &Self::FieldShare) -> Self::FieldShare { self.inner.sub(a, b) }
    fn add_with_public(&mut self, a: &F, b: &Self::FieldShare) -> Self::FieldShare { self.inner.add_with_public(a, b) }
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
    fn mul_vec(&mut self, a: &Self::FieldShareVec, b: &Self::FieldShareVec) -> IoResult<Self::FieldShareVec> { self.inner.mul_vec(a, b) }
    fn promote_to_trivial_share(&self, public_values: F) -> Self::FieldShare { self.inner.promote_to_trivial_share(public_values) }
    fn promote_to_trivial_shares(&self, public_values: &[F]) -> Self::FieldShareVec { self.inner.promote_to_trivial_shares(public_values) }
    fn distribute_powers_and_mul_by_const(&mut self, coeffs: &mut Self::FieldShareVec, g: F, c: F) { self.inner.distribute_powers_and_mul_by_const(coeffs, g, c) }
    fn evaluate_constraint(&mut self, lhs: &[(F, usize)], public_inputs: &[F], private_witness: &Self::FieldShareVec) -> Self::FieldShare { self.inner.evaluate_constraint(lhs, public_inputs, private_witness) }
    /// all rows of one (device-resident) constraint matrix in one launch (shamir.rs:645-663: every party adds the public inputs, one share
    /// component): overrides the row-by-row default (rust/co-groth16-evaluate-constraints.patch)
    fn evaluate_constraints(&mut self, matrix: &[Vec<(F, usize)>], domain_size: usize, public_inputs: &[F], private_witness: &Self::FieldShareVec) -> Self::FieldShareVec {
        let wit = private_witness.clone().get_inner();
        let m = self.gpu.matrix(matrix);
        ShamirPrimeFieldShareVec::new(self.gpu.evaluate_constraints(m, domain_size, public_inputs, -1, &wit, &[]).0)
    }
    fn clone_from_slice(&self, dst: &mut Self::FieldShareVec, src: &Self::FieldShareVec, dst_offset: usize, src_offset: usize, len: usize) { self.inner.clone_from_slice(dst, src, dst_offset, src_offset, len) }
    fn mul_open(&mut self, a: &Self::FieldShare, b: &Self::FieldShare) -> IoResult<F> { self.inner.mul_open(a, b) }
    fn mul_open_many(&mut self, a: &[Self::FieldShare], b: &[Self::FieldShare]) -> IoResult<Vec<F>> { self.inner.mul_open_many(a, b) }
}
impl<C: CurveGroup, N: ShamirNetwork> EcMpcProtocol<C> for ShamirHipProtocol<C::ScalarField, N> {
    type PointShare = ShamirPointShare<C>;
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
impl<P: Pairing, N: ShamirNetwork> PairingEcMpcProtocol<P> for ShamirHipProtocol<P::ScalarField, N> {
    fn open_two_points(&mut self, a: &<Self as EcMpcProtocol<P::G1>>::PointShare, b: &<Self as EcMpcProtocol<P::G2>>::PointShare) -> IoResult<(P::G1, P::G2)> {
        PairingEcMpcProtocol::<P>::open_two_points(&mut self.inner, a, b)
    }
}
impl<F: PrimeField, N: ShamirNetwork> FFTProvider<F> for ShamirHipProtocol<F, N> {
    fn fft<D: EvaluationDomain<F>>(&mut self, data: Self::FieldShareVec, domain: &D) -> Self::FieldShareVec {
        let mut v: Vec<F> = data.into_iter().map(|s| s.inner()).collect();
        v.resize(domain.size(), F::zero());
        self.gpu.ntt(&mut [v.as_mut_slice()], domain.group_gen(), false, None);
        ShamirPrimeFieldShareVec::new(v)
    }
    fn fft_in_place<D: EvaluationDomain<F>>(&mut self, data: &mut Self::FieldShareVec, domain: &D) { *data = self.fft(std::mem::take(data), domain); }
    fn ifft<D: EvaluationDomain<F>>(&mut self, data: &Self::FieldShareVec, domain: &D) -> Self::FieldShareVec {
        let mut v: Vec<F> = data.clone().into_iter().map(|s| s.inner()).collect();
        v.resize(domain.size(), F::zero());
        self.gpu.ntt(&mut [v.as_mut_slice()], domain.group_gen(), true, None);
        ShamirPrimeFieldShareVec::new(v)
    }
    fn ifft_in_place<D: EvaluationDomain<F>>(&mut self, data: &mut Self::FieldShareVec, domain: &D) { *data = self.ifft(data, domain); }
    fn evaluate_poly_public(&mut self, poly: Self::FieldShareVec, point: &F) -> Self::FieldShare { self.inner.evaluate_poly_public(poly, point) }
}
impl<Q: SWCurveConfig, N: ShamirNetwork> MSMProvider<Projective<Q>> for ShamirHipProtocol<Q::ScalarField, N> {
    fn msm_public_points(&mut self, points: &[<Projective<Q> as CurveGroup>::Affine], scalars: &Self::FieldShareVec) -> Self::PointShare {
        debug_assert_eq!(points.len(), scalars.get_len());
        let s: Vec<Q::ScalarField> = scalars.clone().into_iter().map(|x| x.inner()).collect();
        ShamirPointShare::new(self.gpu.msm::<Q>(points, &[s.as_slice()]).pop().unwrap())
    }
}
