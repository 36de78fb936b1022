//! `PlainHipDriver`: the non-MPC driver (plain.rs) with transforms, MSMs and element-wise products on the GPU.  `PlainDriver` holds no
//! state, so everything else forwards to a stock instance.
use crate::gpu::Gpu;
use ark_ec::{pairing::Pairing, short_weierstrass::{Projective, SWCurveConfig}, CurveGroup};
use ark_ff::PrimeField;
use ark_poly::EvaluationDomain;
use mpc_core::{
    protocols::plain::PlainDriver,
    traits::{EcMpcProtocol, FFTProvider, MSMProvider, PairingEcMpcProtocol, PrimeFieldMpcProtocol},
};
use std::io::Result as IoResult;

pub struct PlainHipDriver<F: PrimeField> {
    inner: PlainDriver<F>,
    gpu: Gpu,
}
impl<F: PrimeField> PlainHipDriver<F> {
    pub fn new(device: i32) -> eyre::Result<Self> {
        Ok(Self { inner: PlainDriver::default(), gpu: Gpu::new(device)? })
    }
    pub fn gpu(&mut self) -> &mut Gpu {
        &mut self.gpu
    }
}

impl<F: PrimeField> PrimeFieldMpcProtocol<F> for PlainHipDriver<F> {
    type FieldShare = F;
    type FieldShareVec = Vec<F>;
    fn add(&mut self, a: &F, b: &F) -> F { self.inner.add(a, b) }
    fn sub(&mut self, a: &F, b: &F) -> F { self.inner.sub(a, b) }
    fn add_with_public(&mut self, a: &F, b: &F) -> F { self.inner.add_with_public(a, b) }
    fn sub_assign_vec(&mut self, a: &mut Vec<F>, b: &Vec<F>) { self.inner.sub_assign_vec(a, b) }
    fn mul(&mut self, a: &F, b: &F) -> IoResult<F> { self.inner.mul(a, b) }
    fn mul_many(&mut self, a: &[F], b: &[F]) -> IoResult<Vec<F>> { self.inner.mul_many(a, b) }
    fn mul_with_public(&mut self, a: &F, b: &F) -> F { self.inner.mul_with_public(a, b) }
    fn inv(&mut self, a: &F) -> IoResult<F> { self.inner.inv(a) }
    fn inv_many(&mut self, a: &[F]) -> IoResult<Vec<F>> { self.inner.inv_many(a) }
    fn inv_many_in_place(&mut self, a: &mut [F]) -> IoResult<()> { self.inner.inv_many_in_place(a) }
    fn neg(&mut self, a: &F) -> F { self.inner.neg(a) }
    fn neg_vec_in_place(&mut self, a: &mut Vec<F>) { self.inner.neg_vec_in_place(a) }
    fn neg_vec_in_place_limit(&mut self, a: &mut Vec<F>, limit: usize) { self.inner.neg_vec_in_place_limit(a, limit) }
    fn rand(&mut self) -> IoResult<F> { self.inner.rand() }
    fn open(&mut self, a: &F) -> IoResult<F> { self.inner.open(a) }
    fn open_many(&mut self, a: &[F]) -> IoResult<Vec<F>> { self.inner.open_many(a) }
    fn add_vec(&mut self, a: &Vec<F>, b: &Vec<F>) -> Vec<F> { self.inner.add_vec(a, b) }
    /// plain.rs:219-224: element-wise products
    fn mul_vec(&mut self, a: &Vec<F>, b: &Vec<F>) -> IoResult<Vec<F>> { Ok(self.gpu.mul(a, b)) }
    fn promote_to_trivial_share(&self, public_values: F) -> F { self.inner.promote_to_trivial_share(public_values) }
    fn promote_to_trivial_shares(&self, public_values: &[F]) -> Vec<F> { self.inner.promote_to_trivial_shares(public_values) }
    fn distribute_powers_and_mul_by_const(&mut self, coeffs: &mut Vec<F>, g: F, c: F) { self.inner.distribute_powers_and_mul_by_const(coeffs, g, c) }
    fn evaluate_constraint(&mut self, lhs: &[(F, usize)], public_inputs: &[F], private_witness: &Vec<F>) -> F { self.inner.evaluate_constraint(lhs, public_inputs, private_witness) }
    /// all rows of one (device-resident) constraint matrix in one launch: overrides the row-by-row default (rust/co-groth16-evaluate-constraints.patch)
    fn evaluate_constraints(&mut self, matrix: &[Vec<(F, usize)>], domain_size: usize, public_inputs: &[F], private_witness: &Vec<F>) -> Vec<F> {
        let m = self.gpu.matrix(matrix);
        self.gpu.evaluate_constraints(m, domain_size, public_inputs, -1, private_witness, &[]).0
    }
    fn clone_from_slice(&self, dst: &mut Vec<F>, src: &Vec<F>, dst_offset: usize, src_offset: usize, len: usize) { self.inner.clone_from_slice(dst, src, dst_offset, src_offset, len) }
    fn mul_open(&mut self, a: &F, b: &F) -> IoResult<F> { self.inner.mul_open(a, b) }
    fn mul_open_many(&mut self, a: &[F], b: &[F]) -> IoResult<Vec<F>> { self.inner.mul_open_many(a, b) }
}
impl<C: CurveGroup> EcMpcProtocol<C> for PlainHipDriver<C::ScalarField> {
    type PointShare = C;
    fn add_points(&mut self, a: &C, b: &C) -> C { EcMpcProtocol::<C>::add_points(&mut self.inner, a, b) }
    fn sub_points(&mut self, a: &C, b: &C) -> C { EcMpcProtocol::<C>::sub_points(&mut self.inner, a, b) }
    fn add_assign_points(&mut self, a: &mut C, b: &C) { EcMpcProtocol::<C>::add_assign_points(&mut self.inner, a, b) }
    fn sub_assign_points(&mut self, a: &mut C, b: &C) { EcMpcProtocol::<C>::sub_assign_points(&mut self.inner, a, b) }
    fn add_assign_points_public(&mut self, a: &mut C, b: &C) { EcMpcProtocol::<C>::add_assign_points_public(&mut self.inner, a, b) }
    fn sub_assign_points_public(&mut self, a: &mut C, b: &C) { EcMpcProtocol::<C>::sub_assign_points_public(&mut self.inner, a, b) }
    fn add_assign_points_public_affine(&mut self, a: &mut C, b: &C::Affine) { EcMpcProtocol::<C>::add_assign_points_public_affine(&mut self.inner, a, b) }
    fn sub_assign_points_public_affine(&mut self, a: &mut C, b: &C::Affine) { EcMpcProtocol::<C>::sub_assign_points_public_affine(&mut self.inner, a, b) }
    fn scalar_mul_public_point(&mut self, a: &C, b: &C::ScalarField) -> C { EcMpcProtocol::<C>::scalar_mul_public_point(&mut self.inner, a, b) }
    fn scalar_mul_public_scalar(&mut self, a: &C, b: &C::ScalarField) -> C { EcMpcProtocol::<C>::scalar_mul_public_scalar(&mut self.inner, a, b) }
    fn scalar_mul(&mut self, a: &C, b: &C::ScalarField) -> IoResult<C> { EcMpcProtocol::<C>::scalar_mul(&mut self.inner, a, b) }
    fn open_point(&mut self, a: &C) -> IoResult<C> { EcMpcProtocol::<C>::open_point(&mut self.inner, a) }
    fn open_point_many(&mut self, a: &[C]) -> IoResult<Vec<C>> { EcMpcProtocol::<C>::open_point_many(&mut self.inner, a) }
}
impl<P: Pairing> PairingEcMpcProtocol<P> for PlainHipDriver<P::ScalarField> {
    fn open_two_points(&mut self, a: &P::G1, b: &P::G2) -> IoResult<(P::G1, P::G2)> { PairingEcMpcProtocol::<P>::open_two_points(&mut self.inner, a, b) }
}
impl<F: PrimeField> FFTProvider<F> for PlainHipDriver<F> {
    fn fft<D: EvaluationDomain<F>>(&mut self, mut data: Vec<F>, domain: &D) -> Vec<F> { self.fft_in_place(&mut data, domain); data }
    fn fft_in_place<D: EvaluationDomain<F>>(&mut self, data: &mut Vec<F>, domain: &D) {
        assert!(domain.coset_offset().is_one());
        data.resize(domain.size(), F::zero());
        self.gpu.ntt(&mut [data.as_mut_slice()], domain.group_gen(), false, None);
    }
    fn ifft<D: EvaluationDomain<F>>(&mut self, data: &Vec<F>, domain: &D) -> Vec<F> { let mut d = data.clone(); self.ifft_in_place(&mut d, domain); d }
    fn ifft_in_place<D: EvaluationDomain<F>>(&mut self, data: &mut Vec<F>, domain: &D) {
        assert!(domain.coset_offset().is_one());
        data.resize(domain.size(), F::zero());
        self.gpu.ntt(&mut [data.as_mut_slice()], domain.group_gen(), true, None);
    }
    fn evaluate_poly_public(&mut self, poly: Vec<F>, point: &F) -> F { self.inner.evaluate_poly_public(poly, point) }
}
impl<Q: SWCurveConfig> MSMProvider<Projective<Q>> for PlainHipDriver<Q::ScalarField> {
    fn msm_public_points(&mut self, points: &[<Projective<Q> as CurveGroup>::Affine], scalars: &Vec<Q::ScalarField>) -> Projective<Q> {
        self.gpu.msm::<Q>(points, &[scalars.as_slice()]).pop().unwrap()
    }
}
