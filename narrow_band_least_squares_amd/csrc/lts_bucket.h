// Order statistic by bucket refinement: the per-lane state machine of the FAST-LTS kernel for large arrays
// (solve_bucket.inc).  Plain C++ in this header so that the SAME text runs on the device (one start per lane)
// and in a host test (tests/c_caller/bucket_select_test.cpp: lanes emulated one after the other, checked against
// a sort).
//
// Problem: keys k_0..k_{P-1} (the IEEE bit patterns of |r_k| >= 0 — they order like the values, NaN last, as
// np.argsort orders them) and a rank h.  Wanted: a pair (T, m) such that the h-subset of a C-step is
//     { i : k_i < T }  plus the first m of { i : k_i == T } in index order        (stable rank, m = 0 unless tied)
// The keys are never stored: every pass recomputes them and counts how many fall into each of 64 equal bins of the
// current bracket [lo, lo + 64 << shift); the bin that holds the h-th smallest becomes the next bracket (6 bits per
// pass).  A lane is done as soon as a bin EDGE separates the h-th from the (h+1)-th key (then T = that edge, m = 0:
// usually after two or three passes) or when the bins are single key values (shift == 0: T = the value, m = how
// many of the tied keys belong to the subset).
//
// First pass ("clamp"): 64 bins of 2^shift0 centred on a guess of the threshold; bin 0 also collects everything
// below, bin 63 everything above, so any guess is valid — a bad one costs extra passes, never the result.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NBLS_BK_HD __host__ __device__ __forceinline__
#else
#define NBLS_BK_HD inline
#endif

namespace nbls_bucket {

constexpr int kBins = 64;        // bins per pass (6 bits)
constexpr int kDump = 64;        // bin index of keys outside the bracket (a row of the histogram nobody reads)

struct Lane {
    uint64_t lo;      // lower end of the bracket (its low word is zero while shift >= 32)
    int shift;        // bin width 2^shift
    int hrem;         // 1-based rank of the wanted key among the keys >= lo
    int clamp;        // 1 in the first pass
    int done;
    uint64_t T;       // result
    int m;
};

NBLS_BK_HD int clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return x ? __builtin_clzll(x) : 64;
#endif
}

// Start of a selection.  centre_hw: high word of the guessed threshold; shift0 >= 38: bin width of the first pass.
NBLS_BK_HD void init(Lane& s, uint32_t centre_hw, int shift0, int h, bool active) {
    const uint32_t half = 32u << (shift0 - 32);
    const uint32_t lo_hw = centre_hw > half ? centre_hw - half : 0u;
    s.lo = (uint64_t)lo_hw << 32;
    s.shift = shift0;
    s.hrem = h;
    s.clamp = 1;
    s.done = 0;
    s.T = 0;
    s.m = 0;
    if (!active) {                   // a lane without work: every key is "below the bracket" from the second pass on
        s.done = 1;
        s.lo = 0x8000000000000000ull;
        s.shift = 32;
    }
}

// Bin of a key, any state (reference form).
NBLS_BK_HD int bin_of(const Lane& s, uint64_t key) {
    if (s.clamp) {
        if (key < s.lo) return 0;
        const uint64_t d = (key - s.lo) >> s.shift;
        return d > 63 ? 63 : (int)d;
    }
    if (key < s.lo) return kDump;
    const uint64_t d = (key - s.lo) >> s.shift;
    return d > 63 ? kDump : (int)d;
}

// The same from the key's high word alone: valid while shift >= 32 and the low word of lo is zero
// (shift - 32 <= 25 always: a bracket is at most 2^63 wide).
NBLS_BK_HD bool fast_ok(const Lane& s) { return s.shift >= 32 && (uint32_t)s.lo == 0u; }
NBLS_BK_HD int bin_of_hw_clamp(uint32_t lo_hw, int sh, uint32_t hw) {
    int d = (int)(hw - lo_hw) >> sh;             // both words are below 2^31: the signed difference is exact
    d = d < 0 ? 0 : d;
    return d > 63 ? 63 : d;
}
NBLS_BK_HD int bin_of_hw(uint32_t lo_hw, int sh, uint32_t hw) {
    const uint32_t d = (hw - lo_hw) >> sh;       // hw < lo_hw wraps to >= 2^31, >> 25 at most: still >= 64
    return d > 63u ? kDump : (int)d;
}

// After a pass: bsel = the bin that holds the hrem-th key of the bracket, cbelow = keys of the pass in the bins
// before it, cincl = cbelow + keys in bin bsel (cbelow < hrem <= cincl).
NBLS_BK_HD void update(Lane& s, int bsel, int cbelow, int cincl) {
    if (s.done) return;
    const uint64_t new_lo = (s.clamp && bsel == 0) ? 0ull : s.lo + ((uint64_t)bsel << s.shift);
    const uint64_t new_hi = (s.clamp && bsel == 63) ? 0x8000000000000000ull : s.lo + ((uint64_t)(bsel + 1) << s.shift);
    if (cincl == s.hrem) {           // the bin edge separates the h-th from the (h+1)-th key
        s.T = new_hi;
        s.m = 0;
        s.done = 1;
    } else if (!s.clamp && s.shift == 0) {   // single-value bins: ties at the threshold
        s.T = new_lo;
        s.m = s.hrem - cbelow;
        s.done = 1;
    } else {
        s.hrem -= cbelow;
        s.lo = new_lo;
        const uint64_t wm1 = new_hi - new_lo - 1ull;             // bracket width - 1
        const int bits = 64 - clz64(wm1);                        // 0 for a width of 1
        s.shift = bits > 6 ? bits - 6 : 0;
        s.clamp = 0;
    }
    if (s.done) {                    // park the lane: its keys go to the dump row from now on
        s.lo = 0x8000000000000000ull;
        s.shift = 32;
        s.clamp = 0;
    }
}

// Scan of one lane's 64 counts -> (bsel, cbelow, cincl).  count(b) returns the lane's count of bin b.
template <typename CountFn>
NBLS_BK_HD void scan(const CountFn& count, int hrem, int& bsel, int& cbelow, int& cincl) {
    int cum = 0;
    bsel = 0;
    cbelow = 0;
    cincl = 0x7fffffff;
    for (int b = 0; b < kBins; ++b) {
        const int ci = cum + count(b);
        const bool below = ci < hrem;
        bsel += below;
        cbelow = below ? ci : cbelow;
        cincl = below ? cincl : (ci < cincl ? ci : cincl);
        cum = ci;
    }
    if (bsel > 63) bsel = 63;        // (only a parked lane gets here: its counts are all zero)
}

}  // namespace nbls_bucket
