// scalar-field kernels (NTT, pointwise, SpMV) for bn254 (explicit instantiation; see fr_impl.hpp)
#include "fr_impl.hpp"
CG_INSTANTIATE_FR(Bn254Fr)
