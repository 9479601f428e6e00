// Wave-level reductions on the VALU (DPP row permutations + readlane), shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace nbls_wave {

template <int CTRL>
__device__ inline double dpp_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false));
}
__device__ inline double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: after the four steps every
// lane of a 16-lane row holds the row's result; the four rows are combined through SGPRs.
__device__ inline double sum_f64(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ inline double min_f64(double v) {       // NaN lanes are ignored (fmin)
    v = fmin(v, dpp_f64<0xB1>(v));
    v = fmin(v, dpp_f64<0x4E>(v));
    v = fmin(v, dpp_f64<0x141>(v));
    v = fmin(v, dpp_f64<0x140>(v));
    return fmin(fmin(readlane_f64(v, 0), readlane_f64(v, 16)), fmin(readlane_f64(v, 32), readlane_f64(v, 48)));
}

template <int CTRL>
__device__ inline unsigned int dpp_u32(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ inline unsigned int min_u32(unsigned int v) {
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x141>(v));
    v = min(v, dpp_u32<0x140>(v));
    const unsigned int a = (unsigned int)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned int)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned int c = (unsigned int)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}
__device__ inline unsigned int max_u32(unsigned int v) {
    v = max(v, dpp_u32<0xB1>(v));
    v = max(v, dpp_u32<0x4E>(v));
    v = max(v, dpp_u32<0x141>(v));
    v = max(v, dpp_u32<0x140>(v));
    const unsigned int a = (unsigned int)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned int)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned int c = (unsigned int)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

}  // namespace nbls_wave
