// MSM kernels for bls381_g2 (explicit instantiation; see msm_impl.hpp)
#include "msm_impl.hpp"
CG_INSTANTIATE_MSM(Fp2<Bls381Fq>, Bls381Fr)
