// Host test of csrc/lts_bucket.h (the per-lane state machine of the large-array FAST-LTS kernel): the same header
// text that runs one start per lane on the GPU, driven here one "lane" at a time and checked against a sort.
//   g++ -O2 -std=c++17 -I narrow_band_least_squares_amd/csrc tests/c_caller/bucket_select_test.cpp -o bucket_select_test
// Prints "ok <cases> mean_passes <x> max_passes <n>"; exit code 1 on the first mismatch.
#include "lts_bucket.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using nbls_bucket::Lane;

static uint64_t key_of(double r) {
    uint64_t u;
    const double a = std::fabs(r);
    std::memcpy(&u, &a, 8);
    return u & 0x7fffffffffffffffull;
}

struct Stats { long cases = 0, passes = 0; int maxp = 0; };

static bool run_case(const std::vector<uint64_t>& keys, int h, uint32_t centre, int shift0, Stats& st, const char* what) {
    Lane s;
    nbls_bucket::init(s, centre, shift0, h, true);
    int npass = 0;
    while (!s.done) {
        int cnt[nbls_bucket::kBins + 1] = {0};
        const bool fast = nbls_bucket::fast_ok(s);
        for (uint64_t k : keys) {
            const int b = nbls_bucket::bin_of(s, k);
            if (fast) {                                   // the high-word form must agree wherever it is allowed
                const uint32_t hw = (uint32_t)(k >> 32), lo_hw = (uint32_t)(s.lo >> 32);
                const int sh = s.shift - 32;
                const int bf = s.clamp ? nbls_bucket::bin_of_hw_clamp(lo_hw, sh, hw) : nbls_bucket::bin_of_hw(lo_hw, sh, hw);
                if (bf != b) { std::printf("FAIL %s: fast bin %d != %d (shift %d clamp %d)\n", what, bf, b, s.shift, s.clamp); return false; }
                if (sh > 25) { std::printf("FAIL %s: shift %d\n", what, s.shift); return false; }
            }
            ++cnt[b];
        }
        int bsel, cbelow, cincl;
        nbls_bucket::scan([&](int b) { return cnt[b]; }, s.hrem, bsel, cbelow, cincl);
        if (!(cbelow < s.hrem && s.hrem <= cincl)) { std::printf("FAIL %s: scan invariant (%d %d %d)\n", what, cbelow, s.hrem, cincl); return false; }
        nbls_bucket::update(s, bsel, cbelow, cincl);
        if (++npass > 16) { std::printf("FAIL %s: no termination\n", what); return false; }
    }
    long lt = 0, eq = 0;
    for (uint64_t k : keys) { lt += k < s.T; eq += k == s.T; }
    std::vector<uint64_t> srt(keys);
    std::sort(srt.begin(), srt.end());
    const uint64_t hth = srt[h - 1];
    bool ok = (lt + s.m == h) && s.m >= 0 && s.m <= eq;
    if (s.m == 0) ok = ok && (lt == h);
    else ok = ok && (s.T == hth);
    // the subset {k < T} + first m of {k == T} must be the h smallest by stable rank: every key in it <= hth, and it
    // holds every key < hth
    long lt_h = 0;
    for (uint64_t k : keys) lt_h += k < hth;
    ok = ok && (s.m == 0 ? (s.T > hth) : true) && (lt >= lt_h);
    if (!ok) {
        std::printf("FAIL %s: P %zu h %d T %016llx m %d lt %ld eq %ld hth %016llx\n", what, keys.size(), h, (unsigned long long)s.T, s.m, lt, eq,
                    (unsigned long long)hth);
        return false;
    }
    ++st.cases;
    st.passes += npass;
    st.maxp = std::max(st.maxp, npass);
    return true;
}

int main() {
    std::mt19937_64 rng(12345);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::uniform_real_distribution<double> ud(0.0, 1.0);
    Stats all, typical48, typical50;
    const uint32_t one = 0x3ff00000u;
    for (int rep = 0; rep < 6000; ++rep) {
        const int P = 4 + (int)(rng() % 509);
        const int h = 1 + (int)(rng() % (unsigned)(P - 1));
        std::vector<uint64_t> keys(P);
        const int kind = rep % 12;
        const double scale = std::pow(10.0, 6.0 * ud(rng) - 3.0);
        for (int i = 0; i < P; ++i) {
            double v = nd(rng) * scale;
            if (kind == 1 && i % 3 == 0) v = 0.0;                                   // exact zeros (elemental fits)
            if (kind == 2) v = std::ldexp(std::floor(nd(rng) * 4.0), -3);           // heavy ties on a grid
            if (kind == 3 && i % 5 == 0) v = nd(rng) * 1e-17;                       // rounding-level residuals
            if (kind == 4) v = 1.0 + 1e-13 * (double)(rng() % 7);                   // keys that differ in the last bits only
            if (kind == 5 && i % 7 == 0) v = std::nan("");                          // NaN sorts last
            if (kind == 6 && i % 9 == 0) v = INFINITY;
            if (kind == 7) v = 3.0;                                                 // all equal
            if (kind == 8) v = std::ldexp(1.0, (int)(rng() % 2000) - 1000);         // the whole exponent range
            if (kind == 9) v = 5e-324 * (double)(rng() % 5);                        // denormals and zero
            keys[i] = key_of(v);
        }
        const uint32_t guesses[4] = {one, (uint32_t)(key_of(scale) >> 32), (uint32_t)(rng() >> 33), 0u};
        for (uint32_t g : guesses)
            for (int shift0 : {48, 50, 44, 38}) {
                if (!run_case(keys, h, g, shift0, all, "mixed")) return 1;
            }
        // pass statistics of the shape the kernel sees: residuals of a fit, h about P/2, a guess within a factor 2
        if (kind == 0 || kind == 10 || kind == 11) {
            const int hh = P / 2 + 1;
            std::vector<uint64_t> srt(keys);
            std::sort(srt.begin(), srt.end());
            const double f = std::exp2(2.0 * ud(rng) - 1.0);
            double tv;
            std::memcpy(&tv, &srt[hh - 1], 8);
            if (!run_case(keys, hh, (uint32_t)(key_of(tv * f) >> 32), 48, typical48, "typical48")) return 1;
            if (!run_case(keys, hh, one, 50, typical50, "typical50")) return 1;
        }
    }
    std::printf("ok %ld mean_passes %.2f max_passes %d | guess within 2x, shift0 48: %.2f (max %d) | no guess, shift0 50: %.2f (max %d)\n",
                all.cases, (double)all.passes / all.cases, all.maxp, (double)typical48.passes / typical48.cases, typical48.maxp,
                (double)typical50.passes / typical50.cases, typical50.maxp);
    return 0;
}
