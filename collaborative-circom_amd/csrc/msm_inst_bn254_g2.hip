// MSM kernels for bn254_g2 (explicit instantiation; see msm_impl.hpp)
#include "msm_impl.hpp"
CG_INSTANTIATE_MSM(Fp2<Bn254Fq>, Bn254Fr)
