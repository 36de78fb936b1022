// MSM kernels for bn254_g1 (explicit instantiation; see msm_impl.hpp)
#include "msm_impl.hpp"
CG_INSTANTIATE_MSM(Bn254Fq, Bn254Fr)
