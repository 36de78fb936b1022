// Shared host-side helpers of the backend's translation units (error string, HIP error macro, launch geometry).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include "cogroth16_hip.h"

namespace cg {

inline thread_local std::string g_err;
inline int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return ::cg::fail(e_ == hipErrorOutOfMemory ? CG_ERR_OOM : CG_ERR_HIP,           \
                                                std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

constexpr int GRID_CAP = 2048;   // grid-stride kernels: 256 CUs x 8 workgroups
inline int grid_for(size_t n, int block = 256) { size_t g = (n + block - 1) / block; return (int)std::min<size_t>(std::max<size_t>(g, 1), GRID_CAP); }
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
inline int log2_floor(size_t n) { int l = 0; while (((size_t)2 << l) <= n) l++; return l; }

}  // namespace cg
