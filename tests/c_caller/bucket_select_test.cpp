// Host test of csrc/lts_bucket.h (the per-lane state machine of the large-array FAST-LTS kernel): the same header
// text that runs one start per lane on the GPU, driven here one "lane" at a time and checked against a sort.
//   g++ -O2 -std=c++17 -I narrow_band_least_squares_amd/csrc tests/c_caller/bucket_select_test.cpp -o bucket_select_test
// Prints "ok <cases> ..." with pass statistics; exit code 1 on the first mismatch.
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

struct Stats { long cases = 0, passes = 0, gathers = 0; int maxp = 0; };

// One selection.  extra_hist: histogram passes forced on a lane that could already gather (on the GPU a lane waits
// for the slowest lane of its wave).
static bool run_case(const std::vector<uint64_t>& keys, int h, uint32_t centre, int shift0, int extra_hist, Stats& st, const char* what) {
    Lane s;
    nbls_bucket::init(s, centre, shift0, h, true);
    int npass = 0, forced = 0;
    while (!s.done) {
        if (s.gather_ok && forced >= extra_hist) {
            uint64_t g[nbls_bucket::kCap];
            int n = 0;
            for (uint64_t k : keys)
                if (k - s.glo < s.ghi - s.glo) {
                    if (n >= nbls_bucket::kCap) { std::printf("FAIL %s: gather overflow\n", what); return false; }
                    g[n++] = k;
                }
            if (s.h - s.cb < 1 || s.h - s.cb > n) { std::printf("FAIL %s: gather rank %d of %d\n", what, s.h - s.cb, n); return false; }
            nbls_bucket::finish_gather(s, g, n);
            ++st.gathers;
            ++npass;
            break;
        }
        if (s.gather_ok) ++forced;
        int cnt[nbls_bucket::kBins] = {0};
        const bool fast = nbls_bucket::fast_ok(s);
        for (uint64_t k : keys) {
            const int b = nbls_bucket::bin_of(s, k);
            if (fast) {                                   // the high-word form must agree wherever it is allowed
                const uint32_t hw = (uint32_t)(k >> 32), lo_hw = (uint32_t)(s.lo >> 32);
                const int sh = s.shift - 32;
                const int bf = nbls_bucket::bin_of_hw(lo_hw, sh, hw);
                if (bf != b) { std::printf("FAIL %s: fast bin %d != %d (shift %d)\n", what, bf, b, s.shift); return false; }
                if (sh > 31 || lo_hw >= 0x80000000u) { std::printf("FAIL %s: shift %d lo %08x\n", what, s.shift, lo_hw); return false; }
            }
            ++cnt[b];
        }
        int bsel, cbelow, cincl;
        nbls_bucket::scan([&](int b) { return cnt[b]; }, s.h, bsel, cbelow, cincl);
        if (!(cbelow < s.h && s.h <= cincl)) { std::printf("FAIL %s: scan invariant (%d %d %d)\n", what, cbelow, s.h, cincl); return false; }
        nbls_bucket::update(s, bsel, cbelow, cincl);
        if (++npass > 24) { std::printf("FAIL %s: no termination\n", what); return false; }
    }
    long lt = 0, eq = 0;
    for (uint64_t k : keys) { lt += k < s.T; eq += k == s.T; }
    std::vector<uint64_t> srt(keys);
    std::sort(srt.begin(), srt.end());
    const uint64_t hth = srt[h - 1];
    bool ok = (lt + s.m == h) && s.m >= 0 && s.m <= eq;
    if (s.m == 0) ok = ok && (lt == h) && (s.T > hth);
    else ok = ok && (s.T == hth) && (s.m < eq);           // a tie form only for genuine ties across the rank
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

// The guess the kernel makes without a previous threshold: coarse histogram (one binade per bin) of every 4th key.
static uint32_t sample_centre(const std::vector<uint64_t>& keys, int h, int e0) {
    int cnt[64] = {0};
    int ns = 0;
    Lane c;
    nbls_bucket::init(c, 0, 52, 1, true);
    c.lo = (uint64_t)e0 << 52;
    for (size_t i = 0; i < keys.size(); i += 4) { ++cnt[nbls_bucket::bin_of(c, keys[i])]; ++ns; }
    const int hs = std::max(1, (int)(((long)h * ns + (long)keys.size() - 1) / (long)keys.size()));
    int bsel, cbelow, cincl;
    nbls_bucket::scan([&](int b) { return cnt[b]; }, hs, bsel, cbelow, cincl);
    return nbls_bucket::centre_from_sample(e0, bsel, cbelow, cincl, hs);
}

int main() {
    std::mt19937_64 rng(12345);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::uniform_real_distribution<double> ud(0.0, 1.0);
    Stats all, s120, s496, g48;
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
        const uint32_t guesses[5] = {one, (uint32_t)(key_of(scale) >> 32), (uint32_t)(rng() >> 33), 0u, sample_centre(keys, h, 1023 - 48)};
        for (uint32_t g : guesses)
            for (int shift0 : {47, 50, 44, 38})
                for (int extra : {0, 2})
                    if (!run_case(keys, h, g, shift0, extra, all, "mixed")) return 1;
    }
    // pass statistics of the shapes the kernel sees: residuals of a fit (normal x a scale anywhere in 2^-20 .. 2^4),
    // rank about P/2, guess from the sample
    for (int rep = 0; rep < 4000; ++rep) {
        for (int P : {120, 496}) {
            std::vector<uint64_t> keys(P);
            const double scale = std::exp2(24.0 * ud(rng) - 20.0);
            for (int i = 0; i < P; ++i) keys[i] = key_of(nd(rng) * scale);
            const int h = P / 2 + 1;
            if (!run_case(keys, h, sample_centre(keys, h, 1023 - 48), 47, 0, P == 120 ? s120 : s496, "sampled47")) return 1;
            if (P == 120 && !run_case(keys, h, sample_centre(keys, h, 1023 - 48), 48, 0, g48, "sampled48")) return 1;
        }
    }
    std::printf("ok %ld mean_passes %.2f max_passes %d gathers %ld | sampled guess, shift0 47: P 120 %.2f (max %d), P 496 %.2f (max %d) | shift0 48: P 120 %.2f (max %d)\n",
                all.cases, (double)all.passes / all.cases, all.maxp, all.gathers, (double)s120.passes / s120.cases, s120.maxp,
                (double)s496.passes / s496.cases, s496.maxp, (double)g48.passes / g48.cases, g48.maxp);
    return 0;
}
