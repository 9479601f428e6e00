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

}  // namespace nbls_wave
