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

// NumPy's answer for one pair when a window holds NaN or Inf samples (sum of squares not finite), by ONE
// wave (lanes stride the samples; xa, xb in LDS or global memory).  The reference computes
//     cij = np.correlate(a, b, 'full') / sqrt(sum a^2 * sum b^2);  argmax(cij);  cij.max()
// and np.argmax / max treat NaN as the maximum (first NaN wins):
//   * a NaN sample in either window makes the norm NaN -> every lag NaN -> index 0;
//   * else an Inf sample makes the norm Inf: lags whose overlap touches it are Inf/Inf = NaN, the others
//     finite/Inf = 0 -> the first lag that touches an Inf: sample m of `a` is touched by np.correlate
//     indices m .. m+W-1, sample m of `b` by W-1-m .. 2W-2-m -> min(first Inf of a, W-1 - last Inf of b);
//   * the maximum itself is NaN either way.
// (Sums that overflow to Inf without an Inf sample — |x| > 1e150 — fall back to index 0.)
__device__ inline int nonfinite_argmax(const double* xa, const double* xb, int W, int lane) {
    int has_nan = 0, first_a = 0x7fffffff, last_b = -1;
    for (int n = lane; n < W; n += 64) {
        const double va = xa[n], vb = xb[n];
        has_nan |= (va != va) || (vb != vb);
        if (fabs(va) == __builtin_inf() && n < first_a) first_a = n;
        if (fabs(vb) == __builtin_inf()) last_b = n;            // ascending n: the last one stays
    }
    for (int off = 32; off > 0; off >>= 1) {
        has_nan |= __shfl_xor(has_nan, off, 64);
        first_a = min(first_a, __shfl_xor(first_a, off, 64));
        last_b = max(last_b, __shfl_xor(last_b, off, 64));
    }
    if (has_nan) return 0;
    int k = 0x7fffffff;
    if (first_a != 0x7fffffff) k = first_a;
    if (last_b >= 0 && W - 1 - last_b < k) k = W - 1 - last_b;
    return k == 0x7fffffff ? 0 : k;
}
__device__ inline bool finite_f64(double v) { return fabs(v) < __builtin_inf(); }   // false for NaN and +-Inf

// Arg-max order of the reference for one pair: np.argmax(np.correlate(a, b, 'full') / norm) compares the QUOTIENTS, the
// first maximum wins.  v: raw dot products, k: np.correlate index, ss = sum a^2 * sum b^2 (the norm's square: its root is
// taken only in the rare branch that needs the quotients).  Division is monotone,
// so raw values more than a few ulps apart order like their quotients; closer ones may share a quotient, and then the
// SMALLER index wins although its raw value is the smaller one (reachable on periodic inputs).  A total
// order (quotient descending, index ascending): safe in any reduction tree.
__device__ inline bool better_q(double v1, int k1, double v2, int k2, double ss) {
    if (v1 == v2) return k1 < k2;
    const double hi = fmax(fabs(v1), fabs(v2));
    if (fabs(v1 - v2) <= 1.0e-15 * hi) {
        const double nrm = sqrt(ss);
        const double q1 = v1 / nrm, q2 = v2 / nrm;
        return (q1 > q2) || (q1 == q2 && k1 < k2);
    }
    return v1 > v2;
}

}  // namespace nbls_wave
