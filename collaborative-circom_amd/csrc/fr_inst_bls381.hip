// scalar-field kernels (NTT, pointwise, SpMV) for bls381 (explicit instantiation; see fr_impl.hpp)
#include "fr_impl.hpp"
CG_INSTANTIATE_FR(Bls381Fr)
