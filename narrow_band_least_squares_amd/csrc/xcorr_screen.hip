// Cross-correlation lag pick by INT8 matrix-core screening + exact FP64 verification.
//
// Same contract as xcorr.hip (replaces LsBeam.correlate of lts_array): for every
// (unit, pair) the first arg-max over all 2W-1 lags of np.correlate(x_i, x_j, 'full') and the
// normalised maximum.  The arg-max is found WITHOUT evaluating every lag in FP64:
//
//  1. quantize  each channel window is scaled by its own max|x| to 15-bit integers q (|q| <= 16256)
//               and split into two int8 limbs, q = 128*hi + lo, lo in [-64,63], hi in [-127,127].
//  2. screen    the integer correlation I[d] = sum_n q_i[n+d] q_j[n] is computed EXACTLY for all lags
//               with v_mfma_i32_16x16x64_i8 up to the product of the two low limbs: three limb products,
//               int32 accumulators, I'[d] = 16384*HH + 128*(HL+LH); the dropped LL term is bounded by
//               ||lo_i|| ||lo_j|| (Cauchy-Schwarz).  Because |x*Q/s - q| <= 1/2, the true (scaled)
//               correlation differs from I'[d] by at most
//                   eps = (sum|q_i| + sum|q_j|)/2 + W/4 + ||lo_i|| ||lo_j||
//               for every lag, so the true arg-max is among the lags with I'[d] >= max I' - 2 eps.
//               Each lane keeps the few lags of its accumulator rows that pass that test against its
//               running maximum; an LDS atomic max + filter leaves the candidate list per ordered pair.
//  3. verify    one wave per (unit, pair) evaluates the candidates' correlations in FP64 from the
//               filtered trace and takes the first maximum.  If a candidate buffer overflowed (flat
//               correlation, dead channel) the wave falls back to evaluating every lag in FP64.
//
// The result is the FP64 arg-max (up to FP64 rounding ties, as for any summation order) at a small
// fraction of the FP64 work: the screening runs on the int8 matrix pipe, ~64x the FP64 MFMA rate per
// instruction, x1/3 for the limb products.
//
// Tile mapping (as in xcorr_mfma_kernel): rows r = 16 consecutive lags, columns c = (partner j,
// 16-lag block s), K = 64 samples per MFMA:
//     A[r][k] = q_i[n' + k + r + D0]   (8 byte-shifted copies of the sliding channel in LDS: every lane's
//                                       16-byte fragment is two aligned ds_read_b64 from copy r & 7)
//     B[k][c] = q_j(c)[n' + k - 16 s(c)]
// One workgroup per (unit, two sliding channels, partner group); four waves per sliding channel share
// its lag groups; four tile steps are processed together so that the B fragments are reused.
#include "nbls_internal.h"
#include "wave_ops.h"
#include "screen_kloop.inc"
#include <cstdlib>

namespace {

constexpr int QMAX = 16256;      // 127 * 128
constexpr int KOUT = 26;         // candidates kept per ordered pair
constexpr int NSLOT = 6;         // candidate slots per lane
constexpr int CHDR = 6;          // record header: count, flags (1 interval, 2 list overflow), kk_lo, kk_hi,
                                 //                screening maximum of this direction (f32 bits), theta (f32 bits)
constexpr int CSTRIDE = KOUT + CHDR;   // = 32 ints

typedef int v4i __attribute__((ext_vector_type(4)));

struct QArgs {
    const double* filt;
    int64_t npts_pad;
    int nchans;
    const int32_t* Wb;
    const int32_t* incb;
    const int32_t* unit_off;
    const int32_t* win_off;   // [B] first window of each band's range (window sharding)
    const int32_t* unit_band;
    const int32_t* unit_win;  // [U] window index of every unit (= u - unit_off[band] + win_off[band])
    int u0, nu;               // unit batch
    int WP;                   // padded window bytes (multiple of 16)
    int8_t* qbuf;             // [nu][N][2][WP]
    double* qmeta;            // [nu][N][qms]  ss, L1q, smax, 0, then cum[k] = sum of q^2 over samples < 32k, then sum lo^2
    int qms;                  // doubles per (unit, channel) record = 4 + WP/32 + 4
    // screen
    int S, PFB, CSB, CSA;
    int nsl;                  // sliding channels per workgroup: 2 (8 waves) when the LDS images fit, else 1 (4 waves)
    int npg;                  // partner groups per sliding channel (1: all partners in one workgroup)
    int pgsz;                 // partners per group (<= 16; fewer when the images of all partners do not fit a CU's LDS)
    int ncopy;                // byte-shifted copies of a sliding channel in LDS: 8 (two aligned 8-byte reads per fragment), or 4 with
                              // dword-granular addressing on top (long windows: half the LDS per sample, four 4-byte reads)
    int Wuni;                 // the window length when all bands share it (saves two dependent loads), else 0
    int tab_lds;              // energy tables of the pruning test staged in LDS (else read from qmeta when needed)
    unsigned int tab_inv;     // ceil(2^32 / (WP/32 + 2)): division of a table index by the row length as one v_mul_hi
    int cxx;                  // experiment (option "screen_cxx"): the compiler-scheduled K loop also where a hand-scheduled one exists
    int seed;                 // developer experiment: running maxima seeded from the previous pass's candidate records
    int pretest;              // epilogue: integer pre-test on the accumulators before they are converted (option "screen_pretest")
    int dyn;                  // lag groups dealt dynamically to the waves of a sliding channel (else fixed snake order)
    unsigned long long boffp0, boffp1;   // per-channel LDS skew in 16-byte slots (bank-conflict-free B reads): 4 bits per channel,
                              // packed so that a per-lane lookup is a select and a shift, not a load from the argument block
                              // (two named words: indexing an array member by a lane value is such a load)
    unsigned long long* stamps; // developer: per-workgroup s_memtime stamps (NBLS_SCREEN_STAMPS=1), else NULL
    int ablate;               // developer timing switch (NBLS_ABLATE): 1 no K loop, 2 no staging, 4 no epilogue
    int32_t* cand;            // [nu][N][N][CSTRIDE]: count, overflow, kk...
    // verify
    int npairs;
    const int32_t* pair;
    int vector_len;
    int32_t* lag;
    double* cmax;
};

// XCD-aware item order of the streaming kernels (quantize, verify).  Workgroups are dealt round-robin to the 8
// XCDs (blockIdx % 8), each with its own L2.  Consecutive windows of a band overlap (by half at the usual 50 %):
// every XCD therefore takes ONE contiguous range of the items, ordered so that neighbours share samples, and
// walks it in dispatch order — the second reader of an overlap finds it in that XCD's L2 instead of fetching it
// from HBM again.  -> item index, or -1 for the padding at the end of an XCD's range.
__device__ inline int xcd_item(int block, int per_block, int sub, int total) {
    const int xcd = block & 7, slot = block >> 3;
    const int share = (total + 7) >> 3;
    const int it = xcd * share + slot * per_block + sub;
    return (slot * per_block + sub < share && it < total) ? it : -1;
}
__host__ inline int xcd_grid(int per_block, int total) {
    const int share = (total + 7) >> 3;
    return 8 * ((share + per_block - 1) / per_block);
}

// ------------------------------------------------------------------ 1. quantize
// One wave per (unit, channel).  The window is read once from global memory with lane-contiguous
// loads into the wave's LDS slab (max |x| and sum x^2 on the way); each lane then quantises runs of 8
// consecutive samples from LDS and stores two 8-byte limb groups, so that both the reads and the
// writes of the kernel are coalesced.  The slab is padded by one double per 8 samples (index
// n + n/8): lane g's reads at stride 9 doubles are bank-conflict free.
__device__ inline int qpad(int n) { return n + (n >> 3); }

__global__ __launch_bounds__(256) void quantize_kernel(QArgs a) {
    extern __shared__ double qsm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int N = a.nchans;
    // (channel, unit) in channel-major order: neighbours are consecutive windows of one channel (see xcd_item)
    const int item = xcd_item(blockIdx.x, blockDim.x >> 6, wv, a.nu * N);     // 1..4 waves per workgroup (what the slabs leave room for)
    if (item < 0) return;
    const int ch = item / a.nu, ul = item - ch * a.nu;
    const int u = a.u0 + ul;
    const int band = a.unit_band[u];
    const int w = u - a.unit_off[band] + a.win_off[band];   // window index inside the band (global)
    const int W = a.Wb[band];
    const double* src = a.filt + ((int64_t)band * N + ch) * a.npts_pad + (int64_t)w * a.incb[band];
    const int ng8 = a.WP / 8;
    double* sm = qsm + (size_t)wv * (a.WP + ng8 + ng8 + 8);
    double* e8 = sm + a.WP + ng8;                  // [WP/8] energy of each 8-sample group
    double mx = 0.0, ss = 0.0;
    if ((((uintptr_t)src) & 15) == 0) {          // 16-byte loads: two samples per lane per instruction
#pragma unroll 4
        for (int n = 2 * lane; n < a.WP; n += 128) {
            double2 v = make_double2(0.0, 0.0);
            if (n + 1 < W) v = *(const double2*)(src + n);
            else if (n < W) v.x = src[n];
            const int pn = qpad(n);
            sm[pn] = v.x;
            sm[pn + 1] = v.y;
            mx = fmax(mx, fmax(fabs(v.x), fabs(v.y)));
            ss += v.x * v.x + v.y * v.y;
        }
    } else {
        for (int n = lane; n < a.WP; n += 64) {
            const double v = n < W ? src[n] : 0.0;
            sm[qpad(n)] = v;
            mx = fmax(mx, fabs(v));
            ss += v * v;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmax(mx, __shfl_xor(mx, off, 64));
        ss += __shfl_xor(ss, off, 64);
    }
    const double scale = (mx > 0.0 && mx < __builtin_inf()) ? (double)QMAX / mx : 0.0;
    int8_t* qh = a.qbuf + ((int64_t)ul * N + ch) * 2 * a.WP;
    int8_t* ql = qh + a.WP;
    long long l1 = 0;
    int l2lo = 0;                                  // sum of lo^2 (<= 4096 per sample)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int g = lane; g < ng8; g += 64) {
        unsigned int ph[2] = {0, 0}, pl[2] = {0, 0};
        long long eg = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            double xq = sm[g * 9 + e] * scale;
            xq = (xq == xq) ? xq : 0.0;          // NaN sample, or Inf * 0: quantised as 0 (verify applies NumPy's rule)
            int q = (int)rint(xq);
            q = q > QMAX ? QMAX : (q < -QMAX ? -QMAX : q);
            const int lo = ((q + 64) & 127) - 64;
            const int hi = (q - lo) >> 7;
            l1 += q < 0 ? -q : q;
            l2lo += lo * lo;
            eg += (long long)q * q;
            ph[e >> 2] |= (unsigned int)(hi & 0xff) << (8 * (e & 3));
            pl[e >> 2] |= (unsigned int)(lo & 0xff) << (8 * (e & 3));
        }
        *(uint2*)(qh + g * 8) = make_uint2(ph[0], ph[1]);
        *(uint2*)(ql + g * 8) = make_uint2(pl[0], pl[1]);
        e8[g] = (double)eg;
    }
    for (int off = 32; off > 0; off >>= 1) { l1 += __shfl_xor(l1, off, 64); l2lo += __shfl_xor(l2lo, off, 64); }
    double* m = a.qmeta + ((int64_t)ul * N + ch) * a.qms;
    if (lane == 0) {
        m[0] = ss;
        m[1] = (double)l1;
        m[2] = mx;
        m[3] = 0.0;
        m[6 + a.WP / 32] = (double)l2lo;           // after cum[0 .. WP/32+1]
    }
    // cumulative energy of the quantised window at 32-sample granularity (exact integers in double):
    // cum[k] = sum_{n < 32k} q[n]^2, k = 0..WP/32+1 — the screening kernel bounds whole lag blocks with it.
    // Lane k holds the energy of samples [32k, 32k+32); an inclusive scan over the lanes gives cum[k+1].
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nk = a.WP / 32 + 1;                  // cum[0..nk]
    for (int k0 = 0; k0 <= nk; k0 += 64) {         // (one round up to 2000-sample windows)
        const int k = k0 + lane;
        double pk = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) pk += (4 * k + i < ng8) ? e8[4 * k + i] : 0.0;
        double incl = pk;
        for (int off = 1; off < 64; off <<= 1) {
            const double t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        // carry of the previous rounds: cum[k0] already stored by this wave
        double base = 0.0;
        if (k0 > 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            base = e8[ng8 + 0];
        }
        if (k + 1 <= nk) m[4 + k + 1] = base + incl;
        if (k0 == 0 && lane == 0) m[4] = 0.0;
        if (k0 + 64 <= nk) {                       // keep the running total for the next round
            if (lane == 63) e8[ng8 + 0] = base + incl;
        }
    }
}

// Register form of quantize_kernel for windows up to 64*8*G samples: lane l owns the 8-sample groups
// l, l+64, ... (G of them) and fetches them straight into registers (64-byte runs per lane, 4 KiB per
// wave and pass), so the only LDS use is the small table of group energies.  Without the 12 KB slab per
// wave the occupancy is set by registers, which is what hides the HBM latency of this streaming kernel.
template <int G>
__global__ __launch_bounds__(256) void quantize_reg_kernel(QArgs a) {
    extern __shared__ double qsm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int N = a.nchans;
    const int item = xcd_item(blockIdx.x, 4, wv, a.nu * N);       // channel-major (channel, unit), see xcd_item
    if (item < 0) return;
    const int ch = item / a.nu, ul = item - ch * a.nu;
    const int u = a.u0 + ul;
    const int band = a.unit_band[u];
    const int w = a.unit_win[u];                            // window index inside the band (global)
    const int W = a.Wb[band];
    const double* src = a.filt + ((int64_t)band * N + ch) * a.npts_pad + (int64_t)w * a.incb[band];
    const int ng8 = a.WP / 8;
    double* e8 = qsm + (size_t)wv * (ng8 + 8);     // [WP/8] energy of each 8-sample group (+ scan carry)
    double x[G][8];
    const bool al16 = (((uintptr_t)src) & 15) == 0;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int n0 = 8 * (lane + 64 * i);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[i][e] = 0.0;
        if (n0 + 8 <= W && al16) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const double2 v = *(const double2*)(src + n0 + e);
                x[i][e] = v.x; x[i][e + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (n0 + e < W) x[i][e] = src[n0 + e];
        }
    }
    double mx = 0.0, ss = 0.0;
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { mx = fmax(mx, fabs(x[i][e])); ss += x[i][e] * x[i][e]; }
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    ss = nbls_wave::sum_f64(ss);
    const double scale = (mx > 0.0 && mx < __builtin_inf()) ? (double)QMAX / mx : 0.0;
    int8_t* qh = a.qbuf + ((int64_t)ul * N + ch) * 2 * a.WP;
    int8_t* ql = qh + a.WP;
    long long l1 = 0;
    int l2lo = 0;                                  // sum of lo^2 (<= 4096 per sample)
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int g = lane + 64 * i;
        if (g < ng8) {
            unsigned int ph[2] = {0, 0}, pl[2] = {0, 0};
            long long eg = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                double xq = x[i][e] * scale;
                xq = (xq == xq) ? xq : 0.0;      // NaN sample, or Inf * 0: quantised as 0 (verify applies NumPy's rule)
                int q = (int)rint(xq);
                q = q > QMAX ? QMAX : (q < -QMAX ? -QMAX : q);
                const int lo = ((q + 64) & 127) - 64;
                const int hi = (q - lo) >> 7;
                l1 += q < 0 ? -q : q;
                l2lo += lo * lo;
                eg += (long long)q * q;
                ph[e >> 2] |= (unsigned int)(hi & 0xff) << (8 * (e & 3));
                pl[e >> 2] |= (unsigned int)(lo & 0xff) << (8 * (e & 3));
            }
            *(uint2*)(qh + g * 8) = make_uint2(ph[0], ph[1]);
            *(uint2*)(ql + g * 8) = make_uint2(pl[0], pl[1]);
            e8[g] = (double)eg;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { l1 += __shfl_xor(l1, off, 64); l2lo += __shfl_xor(l2lo, off, 64); }
    double* m = a.qmeta + ((int64_t)ul * N + ch) * a.qms;
    if (lane == 0) {
        m[0] = ss;
        m[1] = (double)l1;
        m[2] = mx;
        m[3] = 0.0;
        m[6 + a.WP / 32] = (double)l2lo;           // after cum[0 .. WP/32+1]
    }
    // cumulative energies as in quantize_kernel
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nk = a.WP / 32 + 1;                  // cum[0..nk]
    for (int k0 = 0; k0 <= nk; k0 += 64) {
        const int k = k0 + lane;
        double pk = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) pk += (4 * k + i < ng8) ? e8[4 * k + i] : 0.0;
        double incl = pk;
        for (int off = 1; off < 64; off <<= 1) {
            const double t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        double base = 0.0;
        if (k0 > 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            base = e8[ng8 + 0];
        }
        if (k + 1 <= nk) m[4 + k + 1] = base + incl;
        if (k0 == 0 && lane == 0) m[4] = 0.0;
        if (k0 + 64 <= nk) {
            if (lane == 63) e8[ng8 + 0] = base + incl;
        }
    }
}

// ------------------------------------------------------------------ 2. screen
__device__ inline int boff_of(const QArgs& a, int ch) {
    const unsigned long long w = (ch & 16) ? a.boffp1 : a.boffp0;
    const unsigned int half = (ch & 8) ? (unsigned int)(w >> 32) : (unsigned int)w;      // 32-bit shifts only
    return (int)((half >> (4 * (ch & 7))) & 15);
}
__device__ inline unsigned int alignbyte(unsigned int hi, unsigned int lo, unsigned int sh) {
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}

// select by a wave-wide lane mask held in scalar registers: one v_cndmask, no per-lane predicate arithmetic
__device__ inline float sel_f32(unsigned long long m, float if_set, float if_clear) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
__device__ inline int sel_i32(unsigned long long m, int if_set, int if_clear) {
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}

// order-preserving float <-> int (for integer atomicMax on LDS)
__device__ inline int f2ord(float f) {
    const int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ inline float ord2f(int o) { return __int_as_float(o ^ ((o >> 31) & 0x7fffffff)); }

#define MFMA_I8(A_, B_, C_) C_ = __builtin_amdgcn_mfma_i32_16x16x64_i8(A_, B_, C_, 0, 0, 0)
// limb products of one tile: HH into its own accumulator, HL and LH (both weighted 128) into a shared one
#define TILE_H(AH, CH, CM) \
    MFMA_I8(AH, bh, CH);   \
    MFMA_I8(AH, bl, CM)
#define TILE_L(AL, CM) MFMA_I8(AL, bh, CM)

// Developer build only (-DNBLS_DEVELOPER): phase stamps and ablation switches; compiled out of the shipped kernels.
#ifdef NBLS_DEVELOPER
__device__ int nbls_dev_realtime = 0;        // screen_stamps = 2: phase stamps from the constant 100 MHz clock (wall time)
__device__ inline void stamp(unsigned long long* p, int slot) {
    if (p) { p[slot] = nbls_dev_realtime ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); }
}
#define NBLS_ABL(bits) (a.ablate & (bits))
#else
__device__ inline void stamp(unsigned long long*, int) {}
#define NBLS_ABL(bits) 0
#endif

constexpr int TB = 4;            // tile steps processed together (they share the B fragments)

// Fragment of the FOUR-copy layout (long windows, QArgs.ncopy == 4): 16 bytes at a 4-byte aligned LDS address.
__device__ inline __attribute__((ext_vector_type(4))) int ld_frag32(const unsigned char* p) {
    typedef const int __attribute__((address_space(3))) * lds_i;     // (not volatile: pairs may become ds_read2_b32)
    const lds_i q = (lds_i)p;
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    return (v4i_){q[0], q[1], q[2], q[3]};
}

typedef int v2i __attribute__((ext_vector_type(2)));
// Two aligned 8-byte LDS reads, volatile so that the compiler cannot fuse reads into ds_read2_b64
// (neither the halves of one fragment nor the halves of two fragments): that form is serviced at half the
// rate with 128-byte bank rows, for which copies r and r+4 of this layout collide (measured: 42 % of
// the kernel's LDS cycles were conflict cycles).  Plain ds_read_b64 is conflict free here.
__device__ inline v4i ld_frag64(const unsigned char* p) {
    typedef const volatile v2i __attribute__((address_space(3))) * lds_v2i;
    const v2i lo = *(lds_v2i)p;
    const v2i hi = *(lds_v2i)(p + 8);
    return (v4i){lo[0], lo[1], hi[0], hi[1]};
}

// One workgroup = one unit x TWO sliding channels (waves 0-3 slide channel 2*cp, waves 4-7 channel
// 2*cp+1): the partner images of all N channels are staged once for both, which halves the staging
// traffic and the per-workgroup fixed costs per unit.
// 8 waves: two sliding channels x 4 waves, or — when the images of all channels do not fit twice — one sliding
// channel x 8 waves (16 elements x 3000 samples: 160 KB, one workgroup per CU; 4 waves left the CU with one wave
// per SIMD: 17.5 -> 13.7 ms at the cfg-4 shape.  16 waves were slower again: 15.5 ms).
// TBV = tile steps per lag group.  4: two workgroups per CU (<= 128 VGPRs).  8: the workgroups that have a CU to
// themselves (9+ elements with long windows: one sliding channel, images of all partners, ~160 KB of LDS) run two
// waves per SIMD and may use 256 VGPRs: eight one-block tiles per group, hand-scheduled K loop that walks the A
// stream once (screen_kloop.inc, NBLS_SCREEN_KLOOP_S1_ASM) — half the LDS bytes per product of the four-tile loop,
// which ran at 83 % of the LDS bandwidth at that shape.
// NCV = byte-shifted copies of a sliding channel: 8, or 4 (long windows, QArgs.ncopy; its own instantiation so that the
// layout is a compile-time constant in the common one — as a run-time value it cost the four-tile kernel 41 spilled VGPRs).
template <int TBV, int NCV = 8>
__global__ __launch_bounds__(512, TBV == 8 ? 2 : 4) void screen_kernel(QArgs a) {
    extern __shared__ unsigned char lds[];
    const int tid = threadIdx.x;
    // (the wave index through readfirstlane: everything derived from it — image rows, channels, row addresses of
    //  the staging loops — is then scalar arithmetic and stays off the vector port)
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.nchans;
    const int NSL = a.nsl;
    const int NCP = (N + NSL - 1) >> (NSL - 1);      // channel pairs (or single channels) per unit (NSL is 1 or 2)
    const int NPG = a.npg;                           // partner groups of <= a.pgsz partners (arrays of > 17 elements, long windows)
    // keep the workgroups of one unit on one XCD (the linear block index % 8 says which blocks share an XCD; the
    // grid is 8*NCP*NPG wide, so that is blockIdx.x % 8) so that the unit's quantised window is fetched into that
    // XCD's L2 once.  Speed only.  Grid rows = groups of eight units: no integer division by run-time values here
    // (they cost ~40 instructions each, and this phase shares the vector port with the other workgroup's products).
    const int grp = blockIdx.y, rem = blockIdx.x;
    const int ul = grp * 8 + (rem & 7);
    int cp = rem >> 3, pg = 0;
    if (NPG > 1) { pg = cp / NCP; cp -= pg * NCP; }
    if (ul >= a.nu) return;
#ifdef NBLS_DEVELOPER
    unsigned long long* stp = (a.stamps && tid == 0) ? a.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
#else
    unsigned long long* const stp = nullptr;
#endif
    stamp(stp, 0);
    const int half = NSL == 2 ? wv >> 2 : 0;         // which sliding channel of the pair (one channel: all waves on it)
    const int ci = NSL * cp + half;
    const bool chan_ok = ci < N;                     // odd N: the last pair has one channel
    const int u = a.u0 + ul;
    // wave-uniform (keeps the K loop scalar); with one window length for all bands no load is needed
    const int W = a.Wuni ? a.Wuni : __builtin_amdgcn_readfirstlane(a.Wb[__builtin_amdgcn_readfirstlane(a.unit_band[u])]);
    const int S = a.S, PFB = a.PFB, CSB = a.CSB, CSA = a.CSA, WP = a.WP;
    const int pgbase = a.pgsz * pg;                  // first partner index of this workgroup's group
    const int NP = (N - 1 - pgbase) < a.pgsz ? (N - 1 - pgbase) : a.pgsz;   // partners handled here

    // LDS: channel images [N][2 limbs][CSB] (channel j skewed by boff[j] sixteen-byte slots so that the
    // B-fragment reads are bank-conflict free for every sliding channel), then per sliding channel the
    // 8 shifted copies [2][2 limbs][8][CSA], then the shared running maxima
    // (with ONE sliding channel per workgroup its own image is not needed: N-1 slots, slot = partner index)
    // images staged: all N channels (two sliding channels, every partner in this workgroup), or — two sliding channels AND
    // partner groups (r04: 17+ elements with short windows) — the group's channels img0 .. img0 + pgsz (one more than a
    // sliding channel's partners: each of the two skips itself), or one sliding channel's NP partners
    const int img0 = (NSL == 2 && NPG > 1) ? pgbase : 0;
    const int nimg = NSL == 2 ? (NPG > 1 ? ((a.pgsz + 1) < (N - pgbase) ? (a.pgsz + 1) : (N - pgbase)) : N) : NP;
    constexpr int NC = NCV;                          // 8, or 4 (long windows)
    unsigned char* Bimg = lds;
    unsigned char* Acop = Bimg + (size_t)nimg * 2 * CSB;
    int* gmax = (int*)(Acop + (size_t)NSL * 2 * NC * CSA);  // [2][16] order-preserving int image of a float
    // merge scalars [2][16] each (outside the images, so they can be initialised before the first barrier)
    int* Mj = gmax + 32;                            // ordered-int image of the pair's maximum
    int* cnt = Mj + 32;
    int* klo = cnt + 32;                            // interval in np.correlate index space
    int* khi = klo + 32;
    int* thS = khi + 32;                            // theta of the pair (f32 bits)
    int* qctr = thS + 32;                           // [2] next lag group of each sliding channel (dynamic dealing), + 2 pad
    // per-channel records of the quantiser (sum x^2, sum |q|, max |x|, sum lo^2), fetched by 4N threads at the top of
    // the kernel and read back after the staging barrier: the thresholds need them for the lane's two channels, and a
    // per-lane fetch from global memory was a second round trip behind the staging one
    double* metaS = (double*)(qctr + 4);            // [N][4]
    // energy tables for the lag-block pruning, f32 rounded UP: tails of the sliding channels, prefix sums of all
    const int NB = WP / 32 + 1;
    float* tailT = (float*)(metaS + 4 * N);         // [NSL][NB + 1]  E_i[32k .. W)
    float* cumT = tailT + NSL * (NB + 1);           // [N][NB + 1]    E_j[0 .. 32k)
    if (tid < 32) {
        gmax[tid] = (int)0x80000000;
        Mj[tid] = (int)0x80000000; cnt[tid] = 0; klo[tid] = 0x7fffffff; khi[tid] = -1; thS[tid] = 0;
        if (tid < 4) qctr[tid] = 0;
#ifdef NBLS_DEVELOPER
        // experiment (option "screen_seed", r04): the running maxima start from the screening maxima the PREVIOUS pass over
        // the same batch left in the candidate records — the best case of any scheme that learns the maximum before the
        // first round of lag groups (a scout pass, a K-split first round): how much would pruning from the start save?
        if (a.seed) {
            const int hs_ = tid >> 4, jj_ = tid & 15, chs_ = NSL * cp + hs_;
            const int pgb_ = a.pgsz * pg;
            if (hs_ < NSL && chs_ < N && pgb_ + jj_ < N - 1 && jj_ < a.pgsz) {
                const int jabs_ = pgb_ + jj_ + (pgb_ + jj_ >= chs_ ? 1 : 0);
                const float m_ = __int_as_float(a.cand[(((int64_t)ul * N + chs_) * N + jabs_) * CSTRIDE + 4]);
                if (m_ == m_ && m_ > -3.0e38f) gmax[tid] = f2ord(m_);
            }
        }
#endif
    }

    // ---- stage all channels' limbs (zero padded front and back); loads are issued in groups of
    //      eight before the LDS stores so that their latencies overlap ----
    const int gB = (CSB - 256) / 16;                 // groups written per image (skew room excluded)
    const int nthr = blockDim.x;
    const int ci0 = NSL * cp;                        // (NSL == 1: the slot of channel ch is ch - (ch > ci0))
    // Staging is laid out by rows so that no integer division is needed: wave w takes image rows
    // (slot, limb) = w, w + nwaves, ... and its lanes the 16-byte groups of the row; the sliding
    // channels' rows (channel, limb) are taken by two waves each.  All global loads of a pass are
    // issued before the LDS stores so that their round trips overlap.
    const int gA = CSA / 8;
    const int nwaves = nthr >> 6;
    const int nrowB = 2 * nimg;
    const int nrowA = 2 * NSL;                       // 4 or 2 rows (channel, limb), nsubA = nwaves / nrowA waves each (2 or 4)
    const int rowA = wv & (nrowA - 1), subA = wv >> (NSL == 2 ? 2 : 1);
    const int strideA = (nwaves / nrowA) * 64;       // groups of a row taken per pass by its waves
    const int hsA = rowA >> 1, limbA = rowA & 1;
    const int chsA = NSL * cp + hsA;
    const bool stage_on = !(NBLS_ABL(2));
    const int8_t* srcA = a.qbuf + (((int64_t)ul * N + (chsA < N ? chsA : 0)) * 2 + limbA) * WP;
    unsigned char* dstA = Acop + ((size_t)(hsA * 2 + limbA) * NC) * CSA;
    double meta_v = 0.0;                              // this thread's entry of the per-channel records (tid < 4N)
    if (tid < 4 * N) {
        const int k = tid & 3;
        meta_v = a.qmeta[((int64_t)ul * N + (tid >> 2)) * a.qms + (k < 3 ? k : 6 + WP / 32)];
    }
#ifdef NBLS_DEVELOPER
    unsigned long long dev_s0 = __builtin_amdgcn_s_memtime(), dev_s1 = 0, dev_s2 = 0;
#endif
    // first pass of the sliding rows (128 groups = 1024 samples per row pass)
    unsigned int asd[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int pa = 0; pa < 2; ++pa) {
        const int g = subA * 64 + lane + pa * strideA;
        if (stage_on && chsA < N && g < gA) {
            if (g * 8 < WP) {
                const uint2 lo2 = *(const uint2*)(srcA + g * 8);
                asd[pa][0] = lo2.x; asd[pa][1] = lo2.y;
            }
            if (g * 8 + 8 < WP) {
                const uint2 hi2 = *(const uint2*)(srcA + g * 8 + 8);
                asd[pa][2] = hi2.x; asd[pa][3] = hi2.y;
            }
        }
    }
    // energy-table entry of this thread (f32 tables for the lag-block pruning): fetched now, stored after the images
    const bool tab_on = a.tab_lds && tid < (NSL + N) * (NB + 1);
    const int tab_row = tab_on ? (int)__umulhi((unsigned)tid, a.tab_inv) : 0, tab_k = tab_on ? tid - tab_row * (NB + 1) : 0;
    double tab_a = 0.0, tab_b = 0.0;
    if (tab_on) {
        const int chs = NSL * cp + tab_row;
        const int chn = tab_row < NSL ? (chs < N ? chs : 0) : tab_row - NSL;
        const double* m = a.qmeta + ((int64_t)ul * N + chn) * a.qms + 4;
        tab_b = m[tab_k];
        if (tab_row < NSL) tab_a = m[NB];
    }
    if (stage_on)
    for (int row0 = wv; row0 < nrowB; row0 += 2 * nwaves) {
        for (int g0 = lane; g0 < gB; g0 += 256) {
            uint4 v[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = row0 + rr * nwaves;
                const int slot = row >> 1, limb = row & 1;
                const int ch = NSL == 2 ? img0 + slot : pgbase + slot + (pgbase + slot >= ci0 ? 1 : 0);
                const int8_t* src = a.qbuf + (((int64_t)ul * N + (row < nrowB ? ch : 0)) * 2 + limb) * WP;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int g = g0 + 64 * q;
                    const int m = g * 16 - PFB;      // sample index of the first byte of this group
                    v[rr][q] = make_uint4(0, 0, 0, 0);
                    if (row < nrowB && g < gB && m >= 0 && m < WP) v[rr][q] = *(const uint4*)(src + m);
                }
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = row0 + rr * nwaves;
                const int slot = row >> 1, limb = row & 1;
                const int ch = NSL == 2 ? img0 + slot : pgbase + slot + (pgbase + slot >= ci0 ? 1 : 0);
                unsigned char* dst = Bimg + ((size_t)slot * 2 + limb) * CSB + 16 * boff_of(a, row < nrowB ? ch : 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int g = g0 + 64 * q;
                    if (row < nrowB && g < gB) *(uint4*)(dst + g * 16) = v[rr][q];
                }
            }
        }
    }
#ifdef NBLS_DEVELOPER
    dev_s1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- 8 byte-shifted copies of each sliding channel: A_r[m] = q_i[m + r], r = 0..7.  A lane's
    //      16-byte fragment at byte offset r is read as two aligned ds_read_b64 from copy r & 7 ----
#pragma unroll
    for (int pa = 0; pa < 2; ++pa) {
        const int g = subA * 64 + lane + pa * strideA;
        if (stage_on && chsA < N && g < gA) {
            const unsigned int* sdw = asd[pa];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int rw = r >> 2, rb = r & 3;
                uint2 o;
                o.x = alignbyte(sdw[rw + 1], sdw[rw + 0], rb);
                o.y = alignbyte(sdw[rw + 2], sdw[rw + 1], rb);
                if (r < NC) *(uint2*)(dstA + (size_t)r * CSA + g * 8) = o;
            }
        }
    }
    for (int g = subA * 64 + lane + 2 * strideA; stage_on && chsA < N && g < gA; g += strideA) {   // long windows
        unsigned int sdw[4] = {0, 0, 0, 0};
        if (g * 8 < WP) {
            const uint2 lo2 = *(const uint2*)(srcA + g * 8);
            sdw[0] = lo2.x; sdw[1] = lo2.y;
        }
        if (g * 8 + 8 < WP) {
            const uint2 hi2 = *(const uint2*)(srcA + g * 8 + 8);
            sdw[2] = hi2.x; sdw[3] = hi2.y;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int rw = r >> 2, rb = r & 3;
            uint2 o;
            o.x = alignbyte(sdw[rw + 1], sdw[rw + 0], rb);
            o.y = alignbyte(sdw[rw + 2], sdw[rw + 1], rb);
            if (r < NC) *(uint2*)(dstA + (size_t)r * CSA + g * 8) = o;
        }
    }
#ifdef NBLS_DEVELOPER
    dev_s2 = __builtin_amdgcn_s_memtime();
#endif
    // ---- energy tables -> LDS (values fetched at the top of the kernel) ----
    if (tab_on) {
        if (tab_row < NSL) tailT[tab_row * (NB + 1) + tab_k] = __double2float_ru(tab_a - tab_b);
        else cumT[(tab_row - NSL) * (NB + 1) + tab_k] = __double2float_ru(tab_b);
    }
    for (int idx = tid + nthr; a.tab_lds && idx < (NSL + N) * (NB + 1); idx += nthr) {      // (more than nthr entries: big arrays)
        const int row = (int)__umulhi((unsigned)idx, a.tab_inv), k = idx - row * (NB + 1);
        if (row < NSL) {
            const int chs = NSL * cp + row;
            const double* m = a.qmeta + ((int64_t)ul * N + (chs < N ? chs : 0)) * a.qms + 4;
            tailT[idx] = __double2float_ru(m[NB] - m[k]);
        } else {
            const double* m = a.qmeta + ((int64_t)ul * N + (row - NSL)) * a.qms + 4;
            cumT[(row - NSL) * (NB + 1) + k] = __double2float_ru(m[k]);
        }
    }
    // ---- lane roles (the pair records are fetched before the staging barrier: one round trip less) ----
    const int c = lane & 15, g = lane >> 4;
    const int ncol = NP * S;
    const bool colvalid = (c < ncol) && chan_ok;
    const int cc = (c < ncol) ? c : 0;               // idle columns mirror column 0 (LDS broadcast)
    int s = 0;                                       // cc / NP and cc % NP for cc < 16, NP >= 2 (three elements: 8 lag blocks): compare chain instead of a division
#pragma unroll
    for (int k = 1; k <= 7; ++k) s += (cc >= k * NP) ? 1 : 0;
    const int jj = cc - s * NP;
    const int cis = chan_ok ? ci : 0;
    const int j = pgbase + jj + (pgbase + jj >= cis ? 1 : 0);
    if (tid < 4 * N) metaS[tid] = meta_v;
    stamp(stp, 1);
    __syncthreads();
    stamp(stp, 2);
    // candidate threshold: 2*eps of the quantisation bound, plus the f32 recombination error (<= 8
    // roundings of 2^-24 relative) of two values of magnitude <= ||q_i|| ||q_j|| (Cauchy-Schwarz)
    float theta, iabs6;
    if (NBLS_ABL(8)) { theta = 1.0f; iabs6 = 0.0f; }
    else {
        const double* mi = metaS + 4 * cis;          // ss, L1, max, sum lo^2
        const double* mj = metaS + 4 * j;
        const double iabs = (mi[2] > 0.0 && mj[2] > 0.0)
                                ? (double)QMAX * (double)QMAX * sqrt(mi[0] * mj[0]) / (mi[2] * mj[2]) : 0.0;
        const double lolo = sqrt(mi[3] * mj[3]);                                // bound of the dropped LL product
        theta = (float)(((mi[1] + mj[1]) * 1.001 + 0.5 * (double)W + 8.0 + 2.0 * lolo) * 1.0001 + 1.0e-6 * iabs);
        iabs6 = (float)(1.0e-6 * iabs);
    }


    const unsigned char* Ah = Acop + ((size_t)(half * 2 + 0) * NC) * CSA;
    const unsigned char* Al = Acop + ((size_t)(half * 2 + 1) * NC) * CSA;
    // row r = lane & 15 reads q_i[. + r]: copy r & 7 at byte offset r & 8 — or, four copies, copy r & 3 at byte offset 4 (r >> 2)
    const int arow = NC == 8 ? (lane & 7) : (lane & 3), aoff = NC == 8 ? (lane & 8) : ((lane & 12));
    const unsigned char* pAh = Ah + (size_t)arow * CSA + 16 * g + aoff;   // + n' + D0
    const unsigned char* pAl = Al + (size_t)arow * CSA + 16 * g + aoff;
    const int jslot = NSL == 2 ? j - img0 : jj;
    const unsigned char* pBh = Bimg + ((size_t)jslot * 2) * CSB + 16 * boff_of(a, j) + PFB + 16 * g - 16 * s;   // + n'
    const unsigned char* pBl = pBh + CSB;
    int* gmaxh = gmax + 16 * half;

    float lmax = -__builtin_inff();
    float sv[NSLOT];
    int sd[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) { sv[q] = -__builtin_inff(); sd[q] = 0; }
    int ilo = 0x7fffffff, ihi = -1;      // lag interval that absorbs what does not fit the slots
    const int step = 16 * S;
    const int ntile = (W + step - 1) / step;
    const int ngrp4 = (ntile + TBV - 1) / TBV;
    // the second sliding channel deals its groups in the opposite order: waves w and w+4 share a SIMD,
    // and the groups' K ranges shrink with p, so every SIMD gets the same matrix-core work
    const int nw = nwaves / NSL;                     // waves per sliding channel: 4, or 8 with one channel per workgroup
    const int wvu = __builtin_amdgcn_readfirstlane(NSL == 2 ? (half ? 3 - (wv & 3) : (wv & 3)) : wv);

// consume NT tiles' accumulators: values in f32 (the int32 limb sums recombined; relative error <=
// 2^-22, covered by theta); publish the group's maximum, then keep what can still be the maximum.
// The vector port is what this phase competes for with the matrix products of the other waves (an MFMA holds
// it for half of its cycles), so the rare work is kept off it: the lags are masked against the window end only in
// the group that reaches it (wave-uniform test), and a value that passes the threshold is placed with wave-wide
// masks in scalar registers — ballot of "slot q free or stale", AND with the lanes still to be placed, two
// v_cndmask per slot tried, out as soon as every lane is placed (usually after the first slot) — instead of a
// per-lane chain through all six slots.  Same placement rule as before: the first free or stale slot, else the interval.
#ifdef NBLS_DEVELOPER
#define DEV_ESTAMP(X) if (lane == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); X += t_ - dev_et; dev_et = t_; }
#else
#define DEV_ESTAMP(X)
#endif
#define SCREEN_EPILOGUE(NT, ACC)                                                                          \
    bool quiet_ = false;                                                                                  \
    if (a.pretest && colvalid) {                                                                          \
        /* integer pre-test: every value of this group is <= 16384 max HH + 128 max M in f32 (conversion, */ \
        /* scaling by powers of two and the rounded sum are monotone); if that bound is below the lane's threshold  */ \
        /* in EVERY lane, the group holds neither a candidate nor a new maximum: no conversions, no slot work       */ \
        int hmx_ = ACC[0][0][0], mmx_ = ACC[0][1][0];                                                     \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                    \
            _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                                         \
                hmx_ = ACC[t][0][reg] > hmx_ ? ACC[t][0][reg] : hmx_;                                     \
                mmx_ = ACC[t][1][reg] > mmx_ ? ACC[t][1][reg] : mmx_;                                     \
            }                                                                                             \
        const float bound_ = 16384.0f * (float)hmx_ + 128.0f * (float)mmx_;                               \
        quiet_ = __builtin_amdgcn_ballot_w64(!(bound_ < lmax - theta)) == 0;                              \
        if (quiet_) { const float gm_ = ord2f(gmaxh[jj]); if (gm_ > lmax) lmax = gm_; }                   \
    }                                                                                                     \
    if (colvalid && !quiet_ && !(NBLS_ABL(4))) {                                                         \
        float v[NT * 4];                                                                                  \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                    \
            _Pragma("unroll") for (int reg = 0; reg < 4; ++reg)                                           \
                v[t * 4 + reg] = 16384.0f * (float)ACC[t][0][reg] + 128.0f * (float)ACC[t][1][reg];       \
        const int dlane = D0 + 16 * s + 4 * g;              /* i32 C/D layout: row = 4*(lane>>4)+reg */   \
        if (D0 + NT * step > W) {                           /* only the group that reaches the window end */ \
            _Pragma("unroll") for (int e = 0; e < NT * 4; ++e)                                            \
                v[e] = dlane + (e >> 2) * step + (e & 3) < W ? v[e] : -__builtin_inff();                  \
        }                                                                                                 \
        float gmx = v[0];                                                                                 \
        _Pragma("unroll") for (int e = 1; e < NT * 4; ++e) gmx = fmaxf(gmx, v[e]);                        \
        DEV_ESTAMP(dev_e1)                                                                                \
        if (gmx > lmax) { lmax = gmx; atomicMax(&gmaxh[jj], f2ord(gmx)); }                                \
        const float gm = ord2f(gmaxh[jj]);                                                                \
        if (gm > lmax) lmax = gm;                                                                         \
        const float thr = lmax - theta;                                                                   \
        DEV_ESTAMP(dev_e2)                                                                                \
        if (__builtin_amdgcn_ballot_w64(gmx >= thr) != 0) {                                               \
            _Pragma("unroll") for (int e = 0; e < NT * 4; ++e) {                                          \
                unsigned long long rem = __builtin_amdgcn_ballot_w64(v[e] >= thr);                        \
                if (rem != 0) {                                                                           \
                    const int d = dlane + (e >> 2) * step + (e & 3);                                      \
                    _Pragma("unroll") for (int q = 0; q < NSLOT; ++q) {                                   \
                        const unsigned long long take = rem & __builtin_amdgcn_ballot_w64(sv[q] < thr);   \
                        sv[q] = sel_f32(take, v[e], sv[q]);                                               \
                        sd[q] = sel_i32(take, d, sd[q]);                                                  \
                        rem &= ~take;                                                                     \
                        if (rem == 0) break;                                                              \
                    }                                                                                     \
                    if (rem != 0 && ((rem >> lane) & 1)) { ilo = d < ilo ? d : ilo; ihi = d > ihi ? d : ihi; } \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
    }

    // ---- tile groups, dealt to the four waves in snake order (they get cheaper with p: K range
    //      W - D0).  A group is skipped when, for every partner, Cauchy-Schwarz on the energies of the
    //      overlapping parts shows that none of its lags can come within theta of the running maximum:
    //      |I[d]| <= sqrt(E_i[d..W) * E_j[0..W-d))  for all d >= D0.  The first round (smallest lags,
    //      where the maximum usually is) establishes the maxima; far lag blocks of coherent windows are
    //      then never computed ----
#ifdef NBLS_DEVELOPER
    unsigned long long dev_kcyc = 0, dev_ksteps = 0, dev_ecyc = 0, dev_e1 = 0, dev_e2 = 0, dev_et = 0;  // cycles inside the K loops of this wave, K steps done
#endif
    if (NBLS_ABL(512)) return;                        // developer: everything up to the staging barrier
    // dynamic dealing: the next lag group of this sliding channel goes to whichever of its four waves is
    // free (ascending p: the small lags, where the maximum usually is, are started first; after pruning
    // the groups cost very different amounts, a fixed deal leaves waves idle at the end).  The first group of a
    // wave is its own index (no draw); the counter starts behind those.
    for (int rnd = 0; chan_ok && !NBLS_ABL(256); ++rnd) {
        int p;
        bool may_prune;
        if (a.dyn) {
            int pdraw = wvu;                         // (waves w and w+4 share a SIMD: the second sliding channel starts from the far end)
            if (rnd > 0) {
                if (lane == 0) pdraw = nw + atomicAdd(&qctr[half], 1);
                pdraw = __builtin_amdgcn_readfirstlane(pdraw);
            }
            p = pdraw;
            if (p >= ngrp4) break;
            may_prune = p >= nw || a.seed;       // (a.seed: developer experiment, see the top of the kernel)
        } else {
            if (rnd * nw >= ngrp4) break;
            p = rnd * nw + ((rnd & 1) ? (nw - 1 - wvu) : wvu);
            if (p >= ngrp4) continue;
            may_prune = rnd > 0;
        }
        const int D0 = TBV * p * step;
        if (may_prune && !NBLS_ABL(32)) {
            bool prunable = true;
            if (colvalid) {
                const int ks = D0 / 32 < NB ? D0 / 32 : NB;
                const int kp = (W - D0 + 31) / 32 < NB ? (W - D0 + 31) / 32 : NB;
                float bound;
                if (a.tab_lds) bound = sqrtf(tailT[half * (NB + 1) + ks] * cumT[j * (NB + 1) + kp]) * 1.000001f;
                else {
                    // (tables read from global memory: the addresses are rebuilt here rather than kept in four
                    //  registers through the K loops)
                    int jv = j, iv = cis;
                    asm volatile("" : "+v"(jv), "+v"(iv));
                    const double* cum_i = a.qmeta + ((int64_t)ul * N + iv) * a.qms + 4;
                    const double* cum_j = a.qmeta + ((int64_t)ul * N + jv) * a.qms + 4;
                    bound = (float)(sqrt((cum_i[NB] - cum_i[ks]) * cum_j[kp]) * (1.0 + 1e-6));
                }
                const float gm = ord2f(gmaxh[jj]);
                prunable = bound * 1.000001f + iabs6 < gm - theta;
            }
            // the bound falls with the lag and the running maxima only rise: once a group cannot hold a candidate,
            // no later group of this sliding channel can
            if (__all(prunable)) { if (a.dyn) break; else continue; }
        }
        v4i acc[TBV][2];
        if constexpr (TBV == 8) {
            // eight one-block tiles (the host picks this instance only for S == 1): see screen_kloop.inc
            typedef const __attribute__((address_space(3))) unsigned char* lds_cp;
            unsigned int va_h = (unsigned int)(uintptr_t)(lds_cp)pAh + D0, vb_h = (unsigned int)(uintptr_t)(lds_cp)pBh;
            asm volatile("" : "+v"(va_h), "+v"(vb_h));
            unsigned int va_l = va_h + 8 * CSA, vb_l = vb_h + CSB;
            const int nst = (W - D0 + 63) >> 6;
            int kcnt;
            if (!(NBLS_ABL(1))) {
                NBLS_SCREEN_KLOOP_S1_ASM(acc[0][0], acc[0][1], acc[1][0], acc[1][1], acc[2][0], acc[2][1], acc[3][0], acc[3][1],
                                         acc[4][0], acc[4][1], acc[5][0], acc[5][1], acc[6][0], acc[6][1], acc[7][0], acc[7][1],
                                         va_h, va_l, vb_h, vb_l, nst, kcnt);
            } else {
#pragma unroll
                for (int t = 0; t < TBV; ++t) { acc[t][0] = (v4i){0, 0, 0, 0}; acc[t][1] = (v4i){0, 0, 0, 0}; }
            }
        } else {
        v4i h0 = {0, 0, 0, 0}, h1 = {0, 0, 0, 0}, h2 = {0, 0, 0, 0}, h3 = {0, 0, 0, 0};   // HH per tile
        v4i m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0}, m3 = {0, 0, 0, 0};   // HL + LH per tile
        const int klen = W - D0;
        const unsigned char* qa_h = pAh + D0;
        const unsigned char* qa_l = pAl + D0;
        if (!(NBLS_ABL(1))) {
        // the partner fragments of the NEXT K step are fetched while this step's products run
        v4i bh = *(const v4i*)(pBh), bl = *(const v4i*)(pBl);
        if (step == 32 && NC == 8 && !a.cxx) {
            // Two lag blocks per tile step (5..8 partners): hand-scheduled K loop, see screen_kloop.inc (generated by
            // tools/gen_screen_kloop.py).  Tile t+2 at K step n and tile t at K step n+1 read the SAME A fragment, so
            // the A stream is walked once and every fragment pair is multiplied by two partner fragments; nothing is
            // copied between registers, every LDS read is requested six products ahead of its use.
            typedef const __attribute__((address_space(3))) unsigned char* lds_cp;
            // (the low-limb addresses are derived here from the high-limb ones, behind an opaque copy, so that they
            //  do not occupy two more registers through the whole kernel; all four are advanced by the loop)
            unsigned int va_h = (unsigned int)(uintptr_t)(lds_cp)pAh + D0, vb_h = (unsigned int)(uintptr_t)(lds_cp)pBh;
            asm volatile("" : "+v"(va_h), "+v"(vb_h));
            unsigned int va_l = va_h + 8 * CSA, vb_l = vb_h + CSB;
            const int nst = (klen + 63) >> 6;
            int kcnt;
#ifdef NBLS_DEVELOPER
            const unsigned long long kt0 = __builtin_amdgcn_s_memtime();
#endif
            NBLS_SCREEN_KLOOP_ASM(h0, m0, h1, m1, h2, m2, h3, m3, va_h, va_l, vb_h, vb_l, nst, kcnt);
#ifdef NBLS_DEVELOPER
            dev_kcyc += __builtin_amdgcn_s_memtime() - kt0; dev_ksteps += nst;
#endif
        } else
        for (int n0 = 0; n0 < klen; n0 += 64) {
            const int nn = n0 + 64 < klen ? n0 + 64 : n0;
            const v4i bhn = *(const v4i*)(pBh + nn);
            const v4i bln = *(const v4i*)(pBl + nn);
#define LDF(P_) (NC == 8 ? ld_frag64(P_) : ld_frag32(P_))
            const v4i a0h = LDF(qa_h + n0);
            const v4i a0l = LDF(qa_l + n0);
            const v4i a1h = LDF(qa_h + n0 + step);
            const v4i a1l = LDF(qa_l + n0 + step);
            const v4i a2h = LDF(qa_h + n0 + 2 * step);
            const v4i a2l = LDF(qa_l + n0 + 2 * step);
            const v4i a3h = LDF(qa_h + n0 + 3 * step);
            const v4i a3l = LDF(qa_l + n0 + 3 * step);
#undef LDF
            TILE_H(a0h, h0, m0);
            TILE_H(a1h, h1, m1);
            TILE_L(a0l, m0);
            TILE_L(a1l, m1);
            TILE_H(a2h, h2, m2);
            TILE_H(a3h, h3, m3);
            TILE_L(a2l, m2);
            TILE_L(a3l, m3);
            bh = bhn; bl = bln;
        }
        }
        acc[0][0] = h0; acc[0][1] = m0; acc[1][0] = h1; acc[1][1] = m1;
        acc[2][0] = h2; acc[2][1] = m2; acc[3][0] = h3; acc[3][1] = m3;
        }
#ifdef NBLS_DEVELOPER
        const unsigned long long et0 = __builtin_amdgcn_s_memtime();
        dev_et = et0;
#endif
        SCREEN_EPILOGUE(TBV, acc)
#ifdef NBLS_DEVELOPER
        dev_ecyc += __builtin_amdgcn_s_memtime() - et0;
#endif
    }
#undef SCREEN_EPILOGUE
    stamp(stp, 3);
#ifdef NBLS_DEVELOPER
    if (stp) { stp[6] = dev_kcyc + (dev_ecyc << 32); stp[7] = dev_e1 + (dev_e2 << 32); (void)dev_ksteps;
               if (NBLS_ABL(2048)) { stp[6] = (dev_s0 - stp[0]) + ((dev_s1 - dev_s0) << 32); stp[7] = (dev_s2 - dev_s1) + ((stp[1] - dev_s2) << 32); } }   // (two fields per word: their means stay separable)
#endif
    if (NBLS_ABL(16)) return;                         // developer: no merge / candidate write
    __syncthreads();       // everyone is done with the A/B images: reuse the LDS head for the merge
    stamp(stp, 4);
    int* lst = (int*)lds;                           // [2][16][KOUT] (aliases the images: they are done with)
    const int hj = 16 * half + jj;
    if (colvalid && lmax > -__builtin_inff()) atomicMax(&Mj[hj], f2ord(lmax));
    if (colvalid && s == 0 && g == 0) thS[hj] = __float_as_int(theta);
    if (colvalid && ihi >= 0) {
        const int k1 = (ci < j) ? (W - 1 + ilo) : (W - 1 - ihi);
        const int k2 = (ci < j) ? (W - 1 + ihi) : (W - 1 - ilo);
        atomicMin(&klo[hj], k1);
        atomicMax(&khi[hj], k2);
    }
    __syncthreads();
    if (colvalid) {
        const float thr = ord2f(Mj[hj]) - theta;
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            if (sv[q] >= thr) {
                const int pos = atomicAdd(&cnt[hj], 1);
                if (pos < KOUT) lst[hj * KOUT + pos] = (ci < j) ? (W - 1 + sd[q]) : (W - 1 - sd[q]);
            }
        }
    }
    __syncthreads();
    for (int item = tid; item < NSL * NP * CSTRIDE; item += nthr) {
        const int hs = item >= NP * CSTRIDE ? 1 : 0, r2 = item - hs * NP * CSTRIDE;
        const int pj = r2 / CSTRIDE, e = r2 % CSTRIDE;
        const int chs = NSL * cp + hs;
        if (chs >= N) continue;
        const int jabs = pgbase + pj + (pgbase + pj >= chs ? 1 : 0);
        int32_t* out = a.cand + (((int64_t)ul * N + chs) * N + jabs) * CSTRIDE;
        const int hq = 16 * hs + pj;
        const int n = cnt[hq];
        int val;
        if (e == 0) val = n < KOUT ? n : KOUT;
        else if (e == 1) val = (khi[hq] >= 0 ? 1 : 0) | (n > KOUT ? 2 : 0);
        else if (e == 2) val = klo[hq];
        else if (e == 3) val = khi[hq];
        else if (e == 4) val = __float_as_int(ord2f(Mj[hq]));
        else if (e == 5) val = thS[hq];
        else val = (e - CHDR < n && e - CHDR < KOUT) ? lst[hq * KOUT + e - CHDR] : 0;
        out[e] = val;
    }
    stamp(stp, 5);
}

// ------------------------------------------------------------------ 3. verify (FP64)
// Exact FP64 correlation values of up to four candidate indices at once (one wave; lanes stride the
// samples, the partner sample is loaded once for the four).  Unused entries carry kk = -1.
// NQ candidate lags of one pair at once: sum_n xa[n + d_q] * xb[n].  Out-of-window samples are read
// from `zero` (a location holding 0.0 in the same memory as xa), which costs one select per sample
// instead of a masked 64-bit value.
// Sum of a double over the 64 lanes, same value in every lane: DPP permutations inside the 16-lane
// rows (VALU latency, no trip through the LDS crossbar), then the four row sums through SGPRs.
__device__ inline double dpp_mov_f64(double v, const int ctrl_sel) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int lo2, hi2;
    switch (ctrl_sel) {
        case 0: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false); break;    // quad_perm [1,0,3,2]
        case 1: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false); break;    // quad_perm [2,3,0,1]
        case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xf, 0xf, false); break;  // row_half_mirror
        default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xf, 0xf, false); break; // row_mirror
    }
    return __hiloint2double(hi2, lo2);
}
__device__ inline double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ inline double wave_sum_f64(double v) {
    v += dpp_mov_f64(v, 0);
    v += dpp_mov_f64(v, 1);
    v += dpp_mov_f64(v, 2);
    v += dpp_mov_f64(v, 3);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// NQ candidate lags of one pair at once: sum_n xa[n + d_q] * xb[n].  Out-of-window samples are read
// from `zero` (a location holding 0.0 in the same memory as xa), which costs one select per sample
// instead of a masked 64-bit value.
template <int NQ>
__device__ inline void wave_dot(const double* xa, const double* xb, const double* zero, int W, const int (&kk)[4],
                                int lane, double (&out)[4]) {
    int d[NQ];
    double acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { d[q] = kk[q] - (W - 1); acc[q] = 0.0; }
#pragma unroll 4
    for (int n = lane; n < W; n += 64) {
        const double vb = xb[n];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = n + d[q];
            const double* pa = ((unsigned)m < (unsigned)W) ? xa + m : zero;
            acc[q] = __builtin_fma(*pa, vb, acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) out[q] = wave_sum_f64(acc[q]);
}

// R CONSECUTIVE candidate lags k0 .. k0+R-1 of one pair at once (long windows of low-frequency bands: the lags within
// the screening bound of the maximum form one run of ~4 around the peak — cfg-4: 3.7 candidates per ordered pair, 97 %
// of them in runs).  The R shifted reads of xa are one coalesced load per 64 samples plus R conflict-free LDS reads of
// a 64 + R - 1 sample chunk (scr: per-wave scratch), instead of R loads: the global-memory verifier is bound by the
// vector-memory path (5 loads per 4 multiply-adds), this form needs 2.
constexpr int VRUN_U = 8;                       // 64-sample steps per trip of wave_dot_run (their loads are in flight together)
constexpr int VRUN_SCR = 64 * VRUN_U + 8;       // doubles of per-wave scratch
template <int R>
__device__ inline void wave_dot_run(const double* xa, const double* xb, int W, int k0, int lane, double* scr, double (&out)[4]) {
    const int d0 = k0 - (W - 1);
    double acc[R];
#pragma unroll
    for (int q = 0; q < R; ++q) acc[q] = 0.0;
    for (int nb = 0; nb < W; nb += 64 * VRUN_U) {
        double vb[VRUN_U], va[VRUN_U];
#pragma unroll
        for (int u = 0; u < VRUN_U; ++u) {
            const int n = nb + 64 * u + lane;
            const int m = n + d0;
            vb[u] = n < W ? xb[n] : 0.0;
            va[u] = ((unsigned)m < (unsigned)W) ? xa[m] : 0.0;
        }
        double vx = 0.0;                                 // the R - 1 samples behind the last step's chunk
        if (lane < R - 1) { const int mx = nb + 64 * VRUN_U + lane + d0; vx = ((unsigned)mx < (unsigned)W) ? xa[mx] : 0.0; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the previous trip's chunk has been read
#pragma unroll
        for (int u = 0; u < VRUN_U; ++u) scr[64 * u + lane] = va[u];
        if (lane < R - 1) scr[64 * VRUN_U + lane] = vx;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);                         // (LDS is in order per wave; this also covers a generic-pointer store)
#pragma unroll
        for (int u = 0; u < VRUN_U; ++u)
#pragma unroll
            for (int q = 0; q < R; ++q) acc[q] = __builtin_fma(scr[64 * u + lane + q], vb[u], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) out[q] = wave_sum_f64(acc[q]);
}

// Verify one pair: xa/xb point at the two channel windows (LDS or global).  The two 32-int candidate
// records are fetched with ONE load per lane (lanes 0-31: i->j record, 32-63: j->i) and read back with
// wave-uniform shuffles, so the wave does not chase a chain of dependent global loads.
// A direction whose screening maximum lies more than theta below the other direction's cannot hold
// the arg-max (same test as inside the screening kernel) and is dropped without any FP64 work.
__device__ inline void verify_pair(const double* xa, const double* xb, const double* zero, int W, int rec,
                                   double ssa, double ssb, int lane, double* best_out, int* bestk_out, double* scr = nullptr) {
    double best = -__builtin_inf();
    int bestk = 0x7fffffff;
    const double nrm = ssa * ssb;                        // (square of the norm: the maxima are compared as the reference compares them, by quotient — better_q)
    if (!nbls_wave::finite_f64(ssa) || !nbls_wave::finite_f64(ssb)) {
        // NaN / Inf samples in a window (gappy trace): the screening saw them as zeros; the answer is NumPy's
        bestk = nbls_wave::nonfinite_argmax(xa, xb, W, lane);
        best = __builtin_nan("");
    } else if (ssa == 0.0 || ssb == 0.0) {   // dead channel: every lag is 0, np.argmax gives index 0
        best = 0.0;
        bestk = 0;
    } else {
        int n1 = __builtin_amdgcn_readlane(rec, 0), f1 = __builtin_amdgcn_readlane(rec, 1);
        int n2 = __builtin_amdgcn_readlane(rec, 32), f2 = __builtin_amdgcn_readlane(rec, 33);
        const float M1 = __int_as_float(__builtin_amdgcn_readlane(rec, 4)), M2 = __int_as_float(__builtin_amdgcn_readlane(rec, 36));
        const float th = __int_as_float(__builtin_amdgcn_readlane(rec, 5));
        const bool any1 = (n1 > 0) || f1, any2 = (n2 > 0) || f2;
        if (any1 && any2 && th > 0.0f) {
            if (M1 - th > M2) { n2 = 0; f2 = 0; }
            else if (M2 - th > M1) { n1 = 0; f1 = 0; }
        }
        const bool full = ((f1 | f2) & 2) || (n1 + n2 == 0 && !((f1 | f2) & 1));
        int r0lo = 0, r0hi = -1, r1lo = 0, r1hi = -1;
        if (full) { r0lo = 0; r0hi = 2 * W - 2; }
        else {
            if (f1 & 1) { r0lo = __builtin_amdgcn_readlane(rec, 2); r0hi = __builtin_amdgcn_readlane(rec, 3); }
            if (f2 & 1) { r1lo = __builtin_amdgcn_readlane(rec, 34); r1hi = __builtin_amdgcn_readlane(rec, 35); }
        }
        const int nlist = full ? 0 : n1 + n2;
        const int nr0 = r0hi >= r0lo ? r0hi - r0lo + 1 : 0;
        const int nr1 = r1hi >= r1lo ? r1hi - r1lo + 1 : 0;
        const int total = nlist + nr0 + nr1;
        if (scr != nullptr && total >= 2) {
            // ---- run form: the listed lags sorted (rank count across the lanes that hold them), then runs of up to
            //      four consecutive lags share their loads; intervals are runs by construction ----
            // (every lane takes part in the shuffle: a permute reads nothing from lanes that are switched off)
            const int src = lane < n1 ? CHDR + lane : 32 + CHDR + lane - n1;
            const int got = __shfl(rec, src & 63, 64);
            const int cv = lane < nlist ? got : 0x7fffffff;
            int rank = 0;
            for (int u = 0; u < nlist; ++u) {
                const int cu = __builtin_amdgcn_readlane(cv, u);
                rank += (cu < cv) || (cu == cv && u < lane);
            }
            const int srt = __builtin_amdgcn_ds_permute((lane < nlist ? rank : lane) * 4, cv);     // srt[rank] = cv
            auto run_from = [&](int k0, int len) {           // lags k0 .. k0+len-1, len <= 4, wave-uniform
                double v[4] = {0.0, 0.0, 0.0, 0.0};
                if (len >= 4) wave_dot_run<4>(xa, xb, W, k0, lane, scr, v);
                else if (len == 3) wave_dot_run<3>(xa, xb, W, k0, lane, scr, v);
                else if (len == 2) wave_dot_run<2>(xa, xb, W, k0, lane, scr, v);
                else { const int kk1[4] = {k0, -1, -1, -1}; wave_dot<1>(xa, xb, zero, W, kk1, lane, v); }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < len && nbls_wave::better_q(v[q], k0 + q, best, bestk, nrm)) { best = v[q]; bestk = k0 + q; }
            };
            for (int p = 0; p < nlist; ) {
                const int k0 = __builtin_amdgcn_readlane(srt, p);
                int len = 1;
                while (len < 4 && p + len < nlist && __builtin_amdgcn_readlane(srt, p + len) == k0 + len) ++len;
                run_from(k0, len);
                p += len;
                while (p < nlist && __builtin_amdgcn_readlane(srt, p) < k0 + len) ++p;      // (a lag listed by both directions)
            }
            for (int lo = r0lo; lo <= r0hi; lo += 4) run_from(lo, r0hi - lo + 1 < 4 ? r0hi - lo + 1 : 4);
            for (int lo = r1lo; lo <= r1hi; lo += 4) run_from(lo, r1hi - lo + 1 < 4 ? r1hi - lo + 1 : 4);
        } else
        for (int q0 = 0; q0 < total; q0 += 4) {
            int kk[4];
            double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = q0 + q;
                int val = -1;
                if (idx < nlist) val = __builtin_amdgcn_readlane(rec, idx < n1 ? CHDR + idx : 32 + CHDR + idx - n1);
                else if (idx < nlist + nr0) val = r0lo + (idx - nlist);
                else if (idx < total) val = r1lo + (idx - nlist - nr0);
                kk[q] = val;
            }
            const int nq = total - q0;                      // wave-uniform
            if (nq >= 4) wave_dot<4>(xa, xb, zero, W, kk, lane, v);
            else if (nq == 3) wave_dot<3>(xa, xb, zero, W, kk, lane, v);
            else if (nq == 2) wave_dot<2>(xa, xb, zero, W, kk, lane, v);
            else wave_dot<1>(xa, xb, zero, W, kk, lane, v);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (kk[q] >= 0 && nbls_wave::better_q(v[q], kk[q], best, bestk, nrm)) { best = v[q]; bestk = kk[q]; }
        }
    }
    *best_out = best;
    *bestk_out = bestk;
}

// Block-per-unit variant: the N channel windows are staged once in LDS (N*W*8 bytes), the waves
// share the unit's pairs.  Used when the windows fit; otherwise verify_kernel reads from global.
__global__ __launch_bounds__(1024) void verify_lds_kernel(QArgs a) {
    extern __shared__ double vsm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwv = blockDim.x >> 6;
    const int P = a.npairs, N = a.nchans;
    const int ul = xcd_item(blockIdx.x, 1, 0, a.nu);        // consecutive windows stay on one XCD (see xcd_item)
    if (ul < 0) return;
    const int u = a.u0 + ul;
    const int band = a.unit_band[u];
    const int w = a.unit_win[u];                            // window index inside the band (global)
    const int W = a.Wb[band];
    const int64_t t0 = (int64_t)w * a.incb[band];
    // issue the candidate-record and norm loads of this wave's pairs first: their latency hides
    // behind the window staging
    int recs[4];
    double ssA[4], ssB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = wv + q * nwv;
        recs[q] = 0; ssA[q] = 0.0; ssB[q] = 0.0;
        if (k < P) {
            const int ci = a.pair[2 * k], cj = a.pair[2 * k + 1];
            const int32_t* l1 = a.cand + (((int64_t)ul * N + ci) * N + cj) * CSTRIDE;
            const int32_t* l2 = a.cand + (((int64_t)ul * N + cj) * N + ci) * CSTRIDE;
            recs[q] = lane < 32 ? l1[lane] : l2[lane - 32];
            ssA[q] = a.qmeta[((int64_t)ul * N + ci) * a.qms];
            ssB[q] = a.qmeta[((int64_t)ul * N + cj) * a.qms];
        }
    }
    for (int ch = wv; ch < N && !(NBLS_ABL(128)); ch += nwv) {
        const double* src = a.filt + ((int64_t)band * N + ch) * a.npts_pad + t0;
        double* dst = vsm + (size_t)ch * W;
        if (((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
#pragma unroll 10
            for (int n = 2 * lane; n + 1 < W; n += 128) *(double2*)(dst + n) = *(const double2*)(src + n);
            if ((W & 1) && lane == 0) dst[W - 1] = src[W - 1];
        } else {
#pragma unroll 4
            for (int n = lane; n < W; n += 64) dst[n] = src[n];
        }
    }
    if (tid == 0) vsm[(size_t)N * W] = 0.0;              // the zero slot of wave_dot
    __syncthreads();
    for (int k0 = (NBLS_ABL(64)) ? P : 0; k0 * nwv + wv < P; k0 += 4) {
        // more than four pairs per wave (big arrays: 62 at 32 elements): the records and norms of the NEXT four pairs are
        // requested before this group's candidates are evaluated (r04: on demand, every pair waited for a global round
        // trip of its own — 17.9 ms of cfg-5's pass)
        int nrecs[4];
        double nssA[4], nssB[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = wv + (k0 + 4 + q) * nwv;
            nrecs[q] = 0; nssA[q] = 0.0; nssB[q] = 0.0;
            if (k < P) {
                const int ci = a.pair[2 * k], cj = a.pair[2 * k + 1];
                const int32_t* l1 = a.cand + (((int64_t)ul * N + ci) * N + cj) * CSTRIDE;
                const int32_t* l2 = a.cand + (((int64_t)ul * N + cj) * N + ci) * CSTRIDE;
                nrecs[q] = lane < 32 ? l1[lane] : l2[lane - 32];
                nssA[q] = a.qmeta[((int64_t)ul * N + ci) * a.qms];
                nssB[q] = a.qmeta[((int64_t)ul * N + cj) * a.qms];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = wv + (k0 + q) * nwv;
            if (k >= P) break;
            const int ci = a.pair[2 * k], cj = a.pair[2 * k + 1];
            const int rec = recs[q];
            const double ssa = ssA[q], ssb = ssB[q];
            double best;
            int bestk;
            verify_pair(vsm + (size_t)ci * W, vsm + (size_t)cj * W, vsm + (size_t)N * W, W, rec, ssa, ssb, lane, &best, &bestk);
            if (lane == 0) {
                const int64_t o = ((int64_t)band * a.vector_len + w) * P + k;
                a.lag[o] = (W - 1) - bestk;
                a.cmax[o] = best / sqrt(ssa * ssb);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { recs[q] = nrecs[q]; ssA[q] = nssA[q]; ssB[q] = nssB[q]; }
    }
}

// Persistent, double-buffered form of verify_lds_kernel (arrays of up to 8 elements whose windows fit LDS twice): ONE
// 16-wave workgroup per CU walks through the units of its XCD's range (see xcd_item).  While the waves evaluate the
// candidates of unit k from one LDS buffer, the windows of unit k+1 stream into the other one by LDS-DMA
// (global_load_lds_dwordx4: no registers, no ds_write) and its candidate records into registers; the (band, window)
// of unit k+3 is fetched at the top of the step, BEFORE the copies are queued — vector loads return in order, so a
// wait for anything queued after the copies would wait for the copies too (the first version of this kernel fetched
// them afterwards and ran no faster than the block-per-unit form) — and the per-band window length / hop come from
// LDS.  The block-per-unit form loads, waits, computes, exits, and its two resident workgroups run in lockstep:
// developer ablations at cfg-3 gave staging 0.85 ms + candidates 1.17 ms + fixed 0.38 ms = the kernel's 2.34 ms,
// i.e. no overlap at all.  Same arithmetic in the same order: identical results.
// LDS: [2][N][wp] doubles (wp even, >= longest window + 2), the zero slot, W[nbands], inc[nbands].  A window that
// starts on an odd sample is copied from one sample earlier (16-byte granules) and read at row + 1.
struct VMeta { int ul, band, w; };

// `vz` is a VGPR holding 0 that the compiler cannot see through: with a visibly wave-uniform address it moves the
// loaded values to SGPRs at once (v_readfirstlane right behind the load = a wait for the load on the spot); this way
// they stay in flight until vmeta_uniform() is applied, a step or two later.
__device__ inline VMeta vmeta_load(const QArgs& a, int xcd, int share, int slot, int vz) {
    VMeta m;
    const int ul = xcd * share + slot;
    m.ul = (slot < share && ul < a.nu) ? ul : -1;
    // unconditional loads from a clamped index: nothing consumes the values in the step that issues them (a select
    // on the loaded value would make the wave wait for it on the spot)
    const int idx = a.u0 + (m.ul >= 0 ? m.ul : a.nu - 1) + vz;
    m.band = a.unit_band[idx];
    m.w = a.unit_win[idx];
    return m;
}
__device__ inline VMeta vmeta_uniform(const VMeta& m) {
    VMeta r;
    r.ul = m.ul;
    r.band = __builtin_amdgcn_readfirstlane(m.band);
    r.w = __builtin_amdgcn_readfirstlane(m.w);
    return r;
}

__global__ __launch_bounds__(1024) void verify_dma_kernel(QArgs a, int wp, int nbands) {
    extern __shared__ double vsm[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int P = a.npairs, N = a.nchans;
    const int xcd = blockIdx.x & 7, nslot = gridDim.x >> 3;
    const int share = (a.nu + 7) >> 3;
    const size_t bufd = (size_t)N * wp;
    double* const zero = vsm + 2 * bufd;
    int* const Wl = (int*)(zero + 1);
    int* const incl = Wl + nbands;
    if (tid == 0) *zero = 0.0;
    for (int b = tid; b < nbands; b += 1024) { Wl[b] = a.Wb[b]; incl[b] = a.incb[b]; }
    // this wave's pairs (the same for every unit)
    int pci[2], pcj[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = wv + 16 * q;
        pci[q] = k < P ? a.pair[2 * k] : 0;
        pcj[q] = k < P ? a.pair[2 * k + 1] : 0;
    }
    int slot = blockIdx.x >> 3;
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    VMeta mc = vmeta_uniform(vmeta_load(a, xcd, share, slot, vz));                 // unit k: evaluated in this step
    VMeta mn = vmeta_uniform(vmeta_load(a, xcd, share, slot + nslot, vz));         // unit k+1: copied in during this step
    VMeta mf = vmeta_load(a, xcd, share, slot + 2 * nslot, vz);                    // unit k+2 (values still in flight)
    int recs_n[2] = {0, 0}, recs_c[2];
    double ss_n = 0.0, ss_c;
    __syncthreads();                                            // the W / inc tables

// queue the copy of unit M's windows into buffer B, and the loads of its candidate records / norms
#define VERIFY_QUEUE(M, B)                                                                                   \
    if ((M).ul >= 0) {                                                                                       \
        const int W_ = Wl[(M).band];                                                                         \
        const double* src_ = a.filt + (int64_t)(M).band * N * a.npts_pad + (int64_t)(M).w * incl[(M).band]; \
        const int off_ = (int)((((uintptr_t)src_) >> 3) & 1);   /* npts_pad is even: the same for every channel */ \
        const int wt_ = W_ + off_;                                                                           \
        const int npc_ = (wt_ + 127) >> 7;                       /* 1 KiB pieces per row */                  \
        for (int q_ = wv; q_ < N * npc_; q_ += 16) {                                                         \
            const int row_ = q_ / npc_, p_ = q_ - row_ * npc_;                                               \
            const int n_ = 128 * p_ + 2 * lane;                                                              \
            double* dst_ = vsm + (size_t)(B) * bufd + (size_t)row_ * wp + 128 * p_;                          \
            if (n_ < wt_)                                                                                    \
                __builtin_amdgcn_global_load_lds((gptr_t)(src_ - off_ + (int64_t)row_ * a.npts_pad + n_), (lptr_t)dst_, 16, 0, 0); \
        }                                                                                                    \
        _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                   \
            recs_n[q_] = 0;                                                                                  \
            if (wv + 16 * q_ < P) {                                                                          \
                const int32_t* l1 = a.cand + (((int64_t)(M).ul * N + pci[q_]) * N + pcj[q_]) * CSTRIDE;      \
                const int32_t* l2 = a.cand + (((int64_t)(M).ul * N + pcj[q_]) * N + pci[q_]) * CSTRIDE;      \
                recs_n[q_] = lane < 32 ? l1[lane] : l2[lane - 32];                                           \
            }                                                                                                \
        }                                                                                                    \
        ss_n = lane < N ? a.qmeta[((int64_t)(M).ul * N + lane) * a.qms] : 0.0;                               \
    }

    int cur = 0;
    VERIFY_QUEUE(mc, 0)
    __syncthreads();                                        // (waits for the copies as well: vmcnt(0))
    while (mc.ul >= 0) {                                    // workgroup-uniform
        slot += nslot;
        const VMeta mg = vmeta_load(a, xcd, share, slot + 2 * nslot, vz);  // unit k+3: needed two steps from now
#pragma unroll
        for (int q = 0; q < 2; ++q) recs_c[q] = recs_n[q];
        ss_c = ss_n;
        VERIFY_QUEUE(mn, cur ^ 1)
        const int W = Wl[mc.band];
        const double* src = a.filt + (int64_t)mc.band * N * a.npts_pad + (int64_t)mc.w * incl[mc.band];
        const double* buf = vsm + (size_t)cur * bufd + (int)((((uintptr_t)src) >> 3) & 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = wv + 16 * q;
            if (k >= P) break;
            const int ci = pci[q], cj = pcj[q];
            const double ssa = readlane_f64(ss_c, ci), ssb = readlane_f64(ss_c, cj);
            double best;
            int bestk;
            verify_pair(buf + (size_t)ci * wp, buf + (size_t)cj * wp, zero, W, recs_c[q], ssa, ssb, lane, &best, &bestk);
            if (lane == 0) {
                const int64_t o = ((int64_t)mc.band * a.vector_len + mc.w) * P + k;
                a.lag[o] = (W - 1) - bestk;
                a.cmax[o] = best / sqrt(ssa * ssb);
            }
        }
        __syncthreads();                                    // the next unit has landed, this buffer is free
        mc = mn;
        mn = vmeta_uniform(mf);
        mf = mg;
        cur ^= 1;
    }
#undef VERIFY_QUEUE
}

__global__ __launch_bounds__(256) void verify_kernel(QArgs a) {
    __shared__ double vscr[4][VRUN_SCR];            // per wave: a chunk of the sliding window (wave_dot_run)
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int P = a.npairs, N = a.nchans;
    if (item >= a.nu * P) return;
    const int ul = item / P, k = item % P;
    const int u = a.u0 + ul;
    const int band = a.unit_band[u];
    const int w = u - a.unit_off[band] + a.win_off[band];   // window index inside the band (global)
    const int W = a.Wb[band];
    const int64_t t0 = (int64_t)w * a.incb[band];
    const int ci = a.pair[2 * k], cj = a.pair[2 * k + 1];
    const double ssa = a.qmeta[((int64_t)ul * N + ci) * a.qms];
    const double ssb = a.qmeta[((int64_t)ul * N + cj) * a.qms];
    double best;
    int bestk;
    const int32_t* l1 = a.cand + (((int64_t)ul * N + ci) * N + cj) * CSTRIDE;
    const int32_t* l2 = a.cand + (((int64_t)ul * N + cj) * N + ci) * CSTRIDE;
    const int rec = lane < 32 ? l1[lane] : l2[lane - 32];
    verify_pair(a.filt + ((int64_t)band * N + ci) * a.npts_pad + t0, a.filt + ((int64_t)band * N + cj) * a.npts_pad + t0,
                a.qmeta + 3 /* always 0.0 */, W, rec, ssa, ssb, lane, &best, &bestk, vscr[threadIdx.x >> 6]);
    if (lane == 0) {
        const int64_t o = ((int64_t)band * a.vector_len + w) * P + k;
        a.lag[o] = (W - 1) - bestk;
        a.cmax[o] = best / sqrt(ssa * ssb);
    }
}

__global__ void probe_mfma_i8_kernel(const int* a, const int* b, int* out) {
    const int lane = threadIdx.x;
    v4i av = {a[lane * 4], a[lane * 4 + 1], a[lane * 4 + 2], a[lane * 4 + 3]};
    v4i bv = {b[lane * 4], b[lane * 4 + 1], b[lane * 4 + 2], b[lane * 4 + 3]};
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, acc, 0, 0, 0);
    out[lane * 4 + 0] = acc[0];
    out[lane * 4 + 1] = acc[1];
    out[lane * 4 + 2] = acc[2];
    out[lane * 4 + 3] = acc[3];
}

int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace

// Per-channel skew (in 16-byte slots, mod 16) that makes the B-fragment ds_read_b128 of every lane
// group conflict free FOR EVERY sliding channel: two lanes of a group may share a slot only if they
// read the same address.  Lane l reads channel j(l), slot o[j] + g - s; depth-first search over o[]
// (translation fixed by o[0] = 0).  Falls back to o[j] = j (a few 2-way conflicts) if nothing is found.
static bool boff_ok(const int* o, int upto, int N, int S) {
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                      {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    const int NP = N - 1;
    for (int ci = 0; ci < N; ++ci) {
        for (int gi = 0; gi < 4; ++gi) {
            int slot_addr[16];
            for (int q = 0; q < 16; ++q) slot_addr[q] = -1000000;
            for (int q = 0; q < 16; ++q) {
                const int l = groups[gi][q];
                int c = l & 15;
                const int g = l >> 4;
                if (c >= NP * S) c = 0;
                const int jj = c % NP, s = c / NP;
                const int j = jj + (jj >= ci ? 1 : 0);
                if (j > upto) continue;
                const int addr = j * 4096 + o[j] + g - s;             // distinct channels never alias
                const int slot = ((o[j] + g - s) % 16 + 16) % 16;
                if (slot_addr[slot] == -1000000) slot_addr[slot] = addr;
                else if (slot_addr[slot] != addr) return false;
            }
        }
    }
    return true;
}

static bool boff_dfs(int* o, int j, int N, int S, long* budget) {
    if (j == N) return true;
    for (int v = 0; v < (j == 0 ? 1 : 16); ++v) {
        if (--(*budget) < 0) return false;            // bounded search: give up, the caller falls back
        o[j] = v;
        if (boff_ok(o, j, N, S) && boff_dfs(o, j + 1, N, S, budget)) return true;
    }
    return false;
}

// Eligibility + LDS size of the screening path for windows of up to maxW samples.
// *G = partners per workgroup.  All N-1 (at most 16) when their images fit; else the largest group size whose
// images fit next to the sliding channel's eight shifted copies in a CU's 160 KB — the workgroups of a sliding
// channel then split its partners (the mechanism that serves 18..32 elements), e.g. 8 elements x 6000 samples:
// two groups of four; 16 elements x 4500 samples: two groups of eight.  Beyond ~7900 samples even two partners
// do not fit (the copies alone take 16 bytes per sample): the caller falls back to the general correlator.
bool nbls_screen_geometry(const nbls_handle* h, int maxW, int* S, int* PFB, int* CSB, int* CSA, int* WP, size_t* lds, int* nsl, int* G, int* ncopy) {
    const int N = h->nchans;
    if (N < 3 || N > 33 || maxW < 64) return false;
    const int NPc = (N - 1) < 16 ? (N - 1) : 16;     // partners per workgroup when everything fits (more than 16: partner groups)
    *WP = round_up(maxW, 16);
    // first the layout of eight byte-shifted copies per sliding channel (16 bytes of LDS per sample, two aligned 8-byte
    // reads per fragment, hand-scheduled K loops); where not even two partners fit beside them (~7900 samples) FOUR copies
    // with dword-granular addressing on top (8 bytes per sample, four 4-byte reads per fragment, the C++ K loop): ~13 000
    // samples — example.py's WINLEN_1 = 60 s at 200 Hz
    for (int nc = h->opt.screen_nc4 ? 4 : 8; nc >= 4; nc -= 4) {        // (option screen_nc4: the four-copy layout also where eight fit — experiment)
        *ncopy = nc;
        for (int g = NPc; g >= 2; --g) {
            if (nc == 4 && g == 16) continue;            // (S == 1 selects the eight-tile instance, built for eight copies)
            *G = g;
            *S = 16 / g;
            if (*S > 8) break;                           // (the column decode handles up to 8 lag blocks per tile)
            *PFB = 16 * (*S - 1);
            // partner image: PFB + window + read-ahead padding, a whole number of 256-B bank rows, plus one
            // row of room for the per-partner skew
            *CSB = round_up(*PFB + *WP + 192, 256) + 256;
            // K round-up + read-ahead of the last tile of a group (sized for the eight-tile groups of the one-block instance
            // where it may be chosen: S == 1)
            int csa = *WP + 144 + ((*S == 1 ? 8 : TB) - 1) * 16 * (*S);
            csa = round_up(csa, 32);
            while (csa % 64 != 32) csa += 32;                // copy stride == 32 B (mod 64): the 8 copies start 8 banks apart (mod 64), conflict-free ds_read_b64
            *CSA = csa;
            // two sliding channels per workgroup (8 waves, all N images) when two such workgroups fit a CU's
            // LDS, else one sliding channel (4 waves, N-1 images)
            // + running maxima and merge scalars (6 x 32 ints) + the per-channel records (4 doubles each); the f32 energy tables of the pruning test are added by
            // the caller when they still fit (nbls_screen_tables)
            // (partner groups: a workgroup with two sliding channels stages the group's g + 1 channels, not all N)
            const size_t lds2 = (size_t)2 * (N - 1 <= 16 ? N : g + 1) * (*CSB) + (size_t)4 * nc * csa + 6 * 128 + 16 + 32 * N + 64;
            const size_t lds1 = (size_t)2 * g * (*CSB) + (size_t)2 * nc * csa + 6 * 128 + 16 + 32 * N + 64;
            const bool force1 = h->opt.screen_nsl1 != 0;                         // option: one sliding channel per workgroup
            if (nc == 8 && g == NPc && lds2 + (size_t)(2 + N) * (*WP / 32 + 2) * 4 <= 80 * 1024 && !force1) { *nsl = 2; *lds = lds2; }
            else { *nsl = 1; *lds = lds1; }
            if (*lds <= 160 * 1024 && *lds >= 1024) return true;
        }
    }
    return false;
}

// The units [ub, ue) — consecutive bands of ONE window length W (nbls_plan: h->wgroups) — through the screening path.
// *launches counts the unit batches (profiling events are taken from h->bev at 4 * *launches).
hipError_t nbls_launch_xcorr_screen_range(nbls_handle* h, int64_t ub, int64_t ue, int gW, int64_t* launches_io) {
    QArgs a{};
    size_t lds = 0;
    if (!nbls_screen_geometry(h, gW, &a.S, &a.PFB, &a.CSB, &a.CSA, &a.WP, &lds, &a.nsl, &a.pgsz, &a.ncopy)) return hipErrorInvalidValue;
    const int N = h->nchans;
    a.npg = (N - 1 + a.pgsz - 1) / a.pgsz;
    a.Wuni = gW;
    {   // energy tables in LDS when they do not cost occupancy (two workgroups per CU, or still one)
        const size_t tab = (size_t)(a.nsl + N) * (a.WP / 32 + 2) * 4;
        const size_t cap = lds <= 80 * 1024 ? 80 * 1024 : 160 * 1024;
        a.tab_lds = lds + tab <= cap ? 1 : 0;
        if (a.tab_lds) lds += tab;
    }
    a.filt = h->d_filt;
    a.npts_pad = h->npts_pad;
    a.nchans = N;
    a.Wb = h->d_W;
    a.incb = h->d_inc;
    a.unit_off = h->d_unit_off;
    a.win_off = h->d_win_off;
    a.unit_band = h->d_unit_band;
    a.unit_win = h->d_unit_win;
    a.qbuf = h->d_qbuf;
    a.qmeta = h->d_qmeta;
    a.qms = 4 + a.WP / 32 + 4;                     // ss, L1, max, 0, cum[WP/32 + 2], sum lo^2, pad
    a.qms += a.qms & 1;
    a.cand = h->d_cand;
    a.npairs = h->npairs;
    a.pair = h->d_pair;
    a.vector_len = h->vector_len;
    a.lag = h->d_lag;
    a.cmax = h->d_cmax;
    a.dyn = h->opt.screen_static ? 0 : 1;
    a.pretest = h->opt.screen_pretest ? 1 : 0;
    a.seed = h->opt.screen_seed ? 1 : 0;
    a.cxx = h->opt.screen_cxx ? 1 : 0;
    {
        const unsigned long long d = (unsigned long long)(a.WP / 32 + 2);      // exact for every index below 2^32 / d
        a.tab_inv = (unsigned int)(((1ull << 32) + d - 1) / d);
    }
    a.ablate = h->opt.ablate;
    a.stamps = h->opt.screen_stamps ? h->d_stamps : nullptr;
#ifdef NBLS_DEVELOPER
    { const int rt = h->opt.screen_stamps == 2 ? 1 : 0; (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(nbls_dev_realtime), &rt, sizeof(int), 0, hipMemcpyHostToDevice, h->stream); }
#endif
    {
        // solved once per array size (cached); a failed/over-budget search falls back to the linear
        // skew o[jj] = jj, which is correct and at most 2-way conflicted
        // (cached in the HANDLE, keyed by array size and tile shape: handles are driven from several host threads
        //  at once — dist.run_on_handles — and a function-local static would be shared by them)
        if (h->skew_n != N || h->skew_s != a.S) {
            int o[32] = {0};
            long budget = 200000;
            // (partner groups smaller than the array: the solved skew assumes all N-1 partners in one workgroup)
            if (N > 16 || a.pgsz < N - 1 || !boff_dfs(o, 0, N, a.S, &budget))
                for (int q = 0; q < 32; ++q) o[q] = q & 15;   // consecutive partners of a group: distinct slots
            for (int q = 0; q < 32; ++q) h->skew_o[q] = o[q];
            h->skew_n = N;
            h->skew_s = a.S;
        }
        a.boffp0 = a.boffp1 = 0;
        for (int q = 0; q < 16; ++q) a.boffp0 |= (unsigned long long)(h->skew_o[q] & 15) << (4 * q);
        for (int q = 16; q < 32; ++q) a.boffp1 |= (unsigned long long)(h->skew_o[q] & 15) << (4 * (q - 16));
    }
    lds += (size_t)h->opt.screen_pad_kb * 1024;                                                      // developer: occupancy experiment
    // eight-tile instance: one lag block per tile step and a CU per workgroup (two waves per SIMD: 256 VGPRs)
    // (option screen_tb8: the eight-tile instance wherever S == 1, also for workgroups that would fit a CU twice)
    const bool tb8 = a.S == 1 && (lds > 80 * 1024 || h->opt.screen_tb8) && !h->opt.screen_tb4 && a.ncopy == 8;
    const void* skern = tb8 ? (const void*)screen_kernel<8, 8> : (a.ncopy == 4 ? (const void*)screen_kernel<4, 4> : (const void*)screen_kernel<4, 8>);
    hipError_t e = hipFuncSetAttribute(skern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    size_t vlds = ((size_t)N * gW + 2) * sizeof(double);   // + the zero slot
    if (vlds <= 158 * 1024) {
        e = hipFuncSetAttribute((const void*)verify_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vlds);
        if (e != hipSuccess) return e;
    }
    // persistent double-buffered verifier (verify_dma_kernel): up to 8 elements, the unit's windows twice in LDS
    const int vwp = (gW + 3) & ~1;                                       // LDS row stride: even, >= W + 2
    const size_t dlds = ((size_t)2 * N * vwp + 2) * sizeof(double) + (size_t)2 * h->nbands * sizeof(int);
    const bool vdma = N <= 8 && h->npairs <= 32 && dlds <= 160 * 1024 &&
                      (h->npts_pad & 1) == 0;
    if (vdma) {
        e = hipFuncSetAttribute((const void*)verify_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds);
        if (e != hipSuccess) return e;
    }
    int64_t launches = *launches_io;
    // unit batches small enough for the quantised windows to stay in the 256 MiB Infinity Cache; equal batches (a
    // multiple of 8 units, the XCD grouping of the screening grid) instead of full ones plus a remainder: a 200-unit
    // tail batch pays four kernel launches and their drain for next to nothing
    int64_t batch = h->screen_batch;
    {
        const int64_t U = ue - ub;
        const int64_t batch_mb = h->opt.screen_batch_mb > 0 ? h->opt.screen_batch_mb : 192;
        int64_t bw = (int64_t)(batch_mb << 20) / ((int64_t)N * 2 * a.WP);
        if (bw < 64) bw = 64;
        if (bw < batch) batch = bw;                  // (never more than the buffers were sized for)
        if (batch > U) batch = U > 0 ? U : 1;
        if (U > batch) {
            const int64_t nb_ = (U + batch - 1) / batch;
            const int64_t eq = ((U + nb_ - 1) / nb_ + 7) / 8 * 8;
            if (eq < batch) batch = eq;
        }
    }
    const int64_t nbatch = (ue - ub + batch - 1) / batch;
    if (h->prof) {
        while ((int64_t)h->bev.size() < 5 * (launches + nbatch)) {
            hipEvent_t ev;
            if (hipEventCreate(&ev) != hipSuccess) return hipErrorOutOfMemory;
            h->bev.push_back(ev);
        }
    }
    const int64_t kMinSolveUnits = h->opt.solve_min_units > 0 ? h->opt.solve_min_units : 8192;   // units per solve launch and result batch, at least (see below)
    int64_t solve_from = ub;
    for (int64_t u0 = ub; u0 < ue; u0 += batch) {
        a.u0 = (int)u0;
        a.nu = (int)((ue - u0) < batch ? (ue - u0) : batch);
        h->last_batch = a.nu;
        hipEvent_t* ev = h->prof ? &h->bev[5 * launches] : nullptr;
        if (ev) (void)hipEventRecord(ev[0], h->stream);
        {
            const int gpl = (a.WP / 8 + 63) / 64;          // 8-sample groups per lane
            const size_t qlds = (size_t)4 * (a.WP / 8 + 8) * sizeof(double);
            // (one instance per group count: a lane holds 8 G samples in registers, and the registers set how many
            //  waves hide the HBM latency of this streaming kernel)
            if (gpl <= 2)
                hipLaunchKernelGGL((quantize_reg_kernel<2>), dim3(xcd_grid(4, a.nu * N)), dim3(256), qlds, h->stream, a);
            else if (gpl == 3)
                hipLaunchKernelGGL((quantize_reg_kernel<3>), dim3(xcd_grid(4, a.nu * N)), dim3(256), qlds, h->stream, a);
            else if (gpl <= 4)
                hipLaunchKernelGGL((quantize_reg_kernel<4>), dim3(xcd_grid(4, a.nu * N)), dim3(256), qlds, h->stream, a);
            else if (gpl <= 6)
                hipLaunchKernelGGL((quantize_reg_kernel<6>), dim3(xcd_grid(4, a.nu * N)), dim3(256), qlds, h->stream, a);
            else if (gpl <= 8)
                hipLaunchKernelGGL((quantize_reg_kernel<8>), dim3(xcd_grid(4, a.nu * N)), dim3(256), qlds, h->stream, a);
            else {
                // one LDS slab per wave: as many waves per workgroup (<= 4) as fit a CU's 160 KB
                const size_t slab = (size_t)(a.WP + a.WP / 4 + 8) * sizeof(double);
                int nwq = (int)((160 * 1024) / slab);
                nwq = nwq > 4 ? 4 : nwq;
                if (nwq < 1) return hipErrorInvalidValue;
                if (slab * nwq > 48 * 1024) {
                    hipError_t qe = hipFuncSetAttribute((const void*)quantize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(slab * nwq));
                    if (qe != hipSuccess) return qe;
                }
                hipLaunchKernelGGL(quantize_kernel, dim3(xcd_grid(nwq, a.nu * N)), dim3(64 * nwq), slab * nwq, h->stream, a);
            }
        }
        if (ev) (void)hipEventRecord(ev[1], h->stream);
        const int ngrp = (a.nu + 7) / 8;
        // 8 waves: two sliding channels x 4, or one channel x 8
        const dim3 sgrid(8 * ((N + a.nsl - 1) / a.nsl) * a.npg, ngrp);
        if (tb8) hipLaunchKernelGGL((screen_kernel<8, 8>), sgrid, dim3(512), lds, h->stream, a);
        else if (a.ncopy == 4) hipLaunchKernelGGL((screen_kernel<4, 4>), sgrid, dim3(512), lds, h->stream, a);
        else hipLaunchKernelGGL((screen_kernel<4, 8>), sgrid, dim3(512), lds, h->stream, a);
        if (ev) (void)hipEventRecord(ev[2], h->stream);
        if (vdma) {
            const int share = (a.nu + 7) >> 3;
            const int per_xcd = h->num_cus > 0 ? (h->num_cus + 7) / 8 : 32;      // one workgroup per CU
            hipLaunchKernelGGL(verify_dma_kernel, dim3(8 * (share < per_xcd ? share : per_xcd)), dim3(1024), dlds, h->stream, a, vwp, h->nbands);
        } else if (vlds <= 158 * 1024)
            // (many pairs per unit: sixteen waves share them — the workgroup has the CU to itself when its windows fill the LDS)
            hipLaunchKernelGGL(verify_lds_kernel, dim3(xcd_grid(1, a.nu)), dim3(h->npairs > 128 && vlds > 80 * 1024 ? 1024 : 512), vlds, h->stream, a);
        else
            hipLaunchKernelGGL(verify_kernel, dim3((a.nu * h->npairs + 3) / 4), dim3(256), 0, h->stream, a);
        if (ev) (void)hipEventRecord(ev[3], h->stream);
        if (h->fuse_solve && (u0 + a.nu - solve_from >= kMinSolveUnits || u0 + a.nu >= ue)) {
            // the solve of the correlated units right behind them: on the same stream (their rows are complete, and on
            // their way to the host, while the next batch is correlated), or — option "overlap" — on the second stream
            // (VALU-bound) beside the next batch's quantiser / screening kernel (matrix pipe + LDS).  Screening batches
            // are sized in BYTES of quantised windows: with long windows or many elements a batch is a few thousand units
            // (cfg-4: 2 091), under three rounds of the large-array LTS kernel's workgroups — such batches are solved
            // several at a time (r04: the per-batch solves had cost cfg-4's share 37 -> 44.6 ms)
            hipStream_t ss = h->stream;
            if (h->solve_on_stream2) {
                while ((int64_t)h->pev.size() <= launches + 1) {
                    hipEvent_t pe;
                    if (hipEventCreateWithFlags(&pe, hipEventDisableTiming) != hipSuccess) return hipErrorOutOfMemory;
                    h->pev.push_back(pe);
                }
                (void)hipEventRecord(h->pev[launches], h->stream);
                (void)hipStreamWaitEvent(h->stream2, h->pev[launches], 0);
                ss = h->stream2;
            }
            // The host trails the GPU by one result batch: what follows the GPU's last kernel is the copy of the LAST batch's
            // rows and the caller's work on them (the dictionary entries of ~10^4 units: 0.4 ms of a 17 ms call).  The last
            // batch of a streamed pass is therefore cut in two — its final `tail` units are solved and handed over on their
            // own, the rows before them travel (and are worked on) while that small solve runs.
            int64_t tail = h->opt.result_tail_units > 0 ? h->opt.result_tail_units : (h->opt.result_tail_units < 0 ? 0 : 2048);
            const int64_t send = u0 + a.nu;
            if (!(h->stream_results && send >= h->nunits && send - solve_from >= 4 * tail)) tail = 0;
            hipError_t se = nbls_launch_solve_range(h, solve_from, send - tail - solve_from, ss);
            if (se != hipSuccess) return se;
            if (tail) {
                if ((se = nbls_queue_result_batch(h, solve_from, send - tail, ss)) != hipSuccess) return se;
                solve_from = send - tail;
                if ((se = nbls_launch_solve_range(h, solve_from, tail, ss)) != hipSuccess) return se;
            }
            if (ev) (void)hipEventRecord(ev[4], ss);
            if ((se = nbls_queue_result_batch(h, solve_from, send, ss)) != hipSuccess) return se;
            solve_from = send;
        } else if (ev) (void)hipEventRecord(ev[4], h->stream);
        ++launches;
    }
    *launches_io = launches;
    return hipGetLastError();
}

// After the last window group of a pass (nbls_launch_xcorr): the hand-over to a chained pass and the join with the
// solve stream.
hipError_t nbls_xcorr_screen_finish(nbls_handle* h, int64_t launches) {
    // the correlation stage of this pass is queued: a pass chained behind it (nbls_execute_after) may start its own
    // correlation stage here — BEFORE the join below, so that it runs beside this pass's last solve
    if (h->ev_xd) { (void)hipEventRecord(h->ev_xd, h->stream); h->ev_xd_recorded = true; h->ev_xd_by_launcher = true; }
    if (h->fuse_solve) h->solve_done = true;
    if (h->solve_on_stream2) {       // join: everything later on `stream` sees the solves
        while ((int64_t)h->pev.size() <= launches) {
            hipEvent_t pe;
            if (hipEventCreateWithFlags(&pe, hipEventDisableTiming) != hipSuccess) return hipErrorOutOfMemory;
            h->pev.push_back(pe);
        }
        (void)hipEventRecord(h->pev[launches], h->stream2);
        (void)hipStreamWaitEvent(h->stream, h->pev[launches], 0);
    }
    if (h->prof) { h->bev_used = (int)(5 * launches); h->prof_fused = h->fuse_solve; }
    h->tim.xcorr_launches = launches;
    return hipGetLastError();
}

hipError_t nbls_launch_probe_mfma_i8(nbls_handle* h, const int* da, const int* db, int* dout) {
    hipLaunchKernelGGL(probe_mfma_i8_kernel, dim3(1), dim3(64), 0, h->stream, da, db, dout);
    return hipGetLastError();
}
