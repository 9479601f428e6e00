// Order statistic by bucket refinement: the per-lane state machine of the FAST-LTS kernel for large arrays
// (solve_bucket.inc).  Plain C++ in this header so that the SAME text runs on the device (one start per lane)
// and in a host test (tests/c_caller/bucket_select_test.cpp: lanes emulated one after the other, checked against
// a sort).
//
// Problem: keys k_0..k_{P-1} (the IEEE bit patterns of |r_k| >= 0 — they order like the values, NaN last, as
// np.argsort orders them) and a rank h.  Wanted: a pair (T, m) such that the h-subset of a C-step is
//     { i : k_i < T }  plus the first m of { i : k_i == T } in index order        (stable rank, m = 0 unless tied)
// The keys are never stored: every pass recomputes them.
//
//   histogram pass   counts ALL keys into 64 bins of width 2^shift from `lo` up; bin 0 also takes everything below
//                    lo, bin 63 everything from lo + 63 * 2^shift up ("clamp").  Counts are absolute ranks, so the
//                    bin that holds the h-th smallest key is found by a prefix scan against h itself, and
//                      * if the bin's upper edge separates the h-th from the (h+1)-th key, T = that edge: done;
//                      * else the bracket [blo, bhi) that is known to hold the h-th key shrinks to the bin, and the
//                        next pass lays bins 1..62 over it (~6 bits per pass), bins 0 / 63 staying pure "below" /
//                        "above" counters — a first pass around a GUESS may miss (bin 0 or 63): any guess is valid,
//                        a bad one costs passes, never the result;
//                      * single-value bins (shift 0) end the search with the tied value and its take count m.
//   gather pass      once a pure bin holds at most kCap keys, one more pass collects those keys themselves and the
//                    threshold is picked among them directly (no luck needed for the last bits).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NBLS_BK_HD __host__ __device__ __forceinline__
#else
#define NBLS_BK_HD inline
#endif

namespace nbls_bucket {

constexpr int kBins = 64;        // bins per pass
constexpr int kCap = 8;          // keys a gather pass may collect per lane
constexpr uint64_t kTop = 0x8000000000000000ull;   // above every key (keys have the sign bit cleared)

struct Lane {
    uint64_t lo;      // lower edge of bin 1's predecessor: bin b = clamp((key - lo) >> shift, 0, 63)
    int shift;        // bin width 2^shift
    int h;            // 1-based rank wanted (absolute)
    int done;
    uint64_t blo, bhi;   // bracket: the h-th smallest key lies in [blo, bhi)
    // gather proposal of the last update (valid while gather_ok)
    int gather_ok;
    uint64_t glo, ghi;   // the pure bin [glo, ghi) that holds the h-th key and at most kCap keys
    int cb;              // keys below glo
    // result
    uint64_t T;
    int m;
};

NBLS_BK_HD int clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return x ? __builtin_clzll(x) : 64;
#endif
}

// Start of a selection: 62 bins of 2^shift0 centred on a guess (its high word), shift0 >= 38.
NBLS_BK_HD void init(Lane& s, uint32_t centre_hw, int shift0, int h, bool active) {
    const uint32_t half = 32u << (shift0 - 32);
    const uint32_t lo_hw = centre_hw > half ? centre_hw - half : 0u;
    s.lo = (uint64_t)lo_hw << 32;
    s.shift = shift0;
    s.h = h;
    s.done = active ? 0 : 1;
    s.blo = 0;
    s.bhi = kTop;
    s.gather_ok = 0;
    s.glo = s.ghi = 0;
    s.cb = 0;
    s.T = 0;
    s.m = 0;
}

// Bin of a key (reference form, any state).
NBLS_BK_HD int bin_of(const Lane& s, uint64_t key) {
    if (key < s.lo) return 0;
    const uint64_t d = (key - s.lo) >> s.shift;
    return d > 63 ? 63 : (int)d;
}
// The same from the key's high word alone: valid while shift >= 32 and the low word of lo is zero (both high
// words are below 2^31, so the signed difference is exact; shift - 32 <= 26: a bracket is at most 2^63 wide).
NBLS_BK_HD bool fast_ok(const Lane& s) { return s.shift >= 32 && (uint32_t)s.lo == 0u; }
NBLS_BK_HD int bin_of_hw(uint32_t lo_hw, int sh, uint32_t hw) {
    int d = (int)(hw - lo_hw) >> sh;
    d = d < 0 ? 0 : d;
    return d > 63 ? 63 : d;
}

// After a histogram pass: bsel = the bin that holds the h-th smallest key, cbelow = keys in the bins before it,
// cincl = cbelow + keys in bin bsel (cbelow < h <= cincl; all absolute).
NBLS_BK_HD void update(Lane& s, int bsel, int cbelow, int cincl) {
    if (s.done) return;
    const uint64_t edge_lo = s.lo + ((uint64_t)bsel << s.shift);             // meaningful for bsel >= 1 (and for lo == 0)
    const uint64_t edge_hi = s.lo + ((uint64_t)(bsel + 1) << s.shift);       // meaningful for bsel <= 62
    if (bsel <= 62 && cincl == s.h) {            // the bin's upper edge separates the h-th from the (h+1)-th key
        s.T = edge_hi;
        s.m = 0;
        s.done = 1;
        return;
    }
    const bool pure = (bsel >= 1 && bsel <= 62) || (bsel == 0 && s.lo == 0);  // the bin holds no key from outside its own range
    if (pure && s.shift == 0) {                  // single-value bin: ties at the threshold
        s.T = edge_lo;
        s.m = s.h - cbelow;
        s.done = 1;
        return;
    }
    const uint64_t nlo = (bsel == 0) ? s.blo : (edge_lo > s.blo ? edge_lo : s.blo);
    const uint64_t nhi = (bsel == 63) ? s.bhi : (edge_hi < s.bhi ? edge_hi : s.bhi);
    s.blo = nlo;
    s.bhi = nhi;
    s.gather_ok = pure && (cincl - cbelow) <= kCap;
    s.glo = edge_lo;
    s.ghi = edge_hi;
    s.cb = cbelow;
    // next bins: 62 of them over the bracket, bin 0 = everything below it (lo = blo - one bin; no room below: lo = 0)
    const uint64_t wm1 = nhi - nlo - 1ull;                       // bracket width - 1
    int sh = 64 - clz64(wm1) - 6;                                // width <= 64 * 2^sh ...
    sh = sh < 0 ? 0 : sh;
    if ((wm1 >> sh) >= 62ull) ++sh;                              // ... and <= 62 * 2^sh: the smallest such sh
    s.shift = sh;
    const uint64_t one = 1ull << sh;
    s.lo = nlo >= one ? nlo - one : 0ull;
}

// After a gather pass: g[0..n) are the keys of [glo, ghi) (any order), n <= kCap.
NBLS_BK_HD void finish_gather(Lane& s, const uint64_t* g, int n) {
    if (s.done) return;
    const int r = s.h - s.cb;                    // 1-based rank inside the gathered keys
    uint64_t T = 0;
    int lt = 0, eq = 0;
    for (int i = 0; i < kCap; ++i) {
        if (i >= n) continue;
        int l = 0, e = 0;
        for (int j = 0; j < kCap; ++j) {
            if (j >= n) continue;
            l += g[j] < g[i];
            e += g[j] == g[i];
        }
        if (l < r && r <= l + e) { T = g[i]; lt = l; eq = e; }
    }
    if (lt + eq == r) { s.T = T + 1ull; s.m = 0; }       // every key equal to T belongs to the subset: a plain cut
    else { s.T = T; s.m = r - lt; }
    s.done = 1;
}

// Scan of one lane's 64 counts -> (bsel, cbelow, cincl).  count(b) returns the lane's count of bin b.
template <typename CountFn>
NBLS_BK_HD void scan(const CountFn& count, int h, int& bsel, int& cbelow, int& cincl) {
    int cum = 0;
    bsel = 0;
    cbelow = 0;
    cincl = 0x7fffffff;
    for (int b = 0; b < kBins; ++b) {
        const int ci = cum + count(b);
        const bool below = ci < h;
        bsel += below;
        cbelow = below ? ci : cbelow;
        cincl = below ? cincl : (ci < cincl ? ci : cincl);
        cum = ci;
    }
    if (bsel > 63) bsel = 63;        // (only a lane without work gets here: its counts are all zero)
}

// Guess of the threshold's high word from a coarse histogram of a SAMPLE of the keys (bins of one binade:
// shift 52, bin b = biased exponent e0 + b): the sample's quantile bin and the position inside it.
NBLS_BK_HD uint32_t centre_from_sample(int e0, int bsel, int cbelow, int cincl, int hs) {
    const int n = cincl - cbelow;
    uint32_t frac = 0x80000u;                                    // middle of the binade (in units of 2^-20 binade)
    if (n > 0 && cincl != 0x7fffffff) {
        const int num = 2 * (hs - cbelow) - 1;                   // (rank inside the bin - 1/2) / n
        frac = (uint32_t)(((uint64_t)(num < 1 ? 1 : num) << 19) / (uint32_t)n);
        if (frac > 0xfffffu) frac = 0xfffffu;
    }
    int e = e0 + bsel;
    if (e < 0) e = 0;
    if (e > 2046) e = 2046;
    return ((uint32_t)e << 20) + frac;
}

}  // namespace nbls_bucket
