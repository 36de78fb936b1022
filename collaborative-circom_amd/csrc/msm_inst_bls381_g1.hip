// MSM kernels for bls381_g1 (explicit instantiation; see msm_impl.hpp)
#include "msm_impl.hpp"
CG_INSTANTIATE_MSM(Bls381Fq, Bls381Fr)
