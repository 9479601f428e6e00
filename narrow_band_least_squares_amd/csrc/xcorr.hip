// Windowed pairwise full-lag cross-correlation, lag pick and correlation maximum for every
// (band, window, pair) of the plan in one batched launch.
//
// Replaces LsBeam.correlate of lts_array (called through ltsva at
// narrow_band_least_squares.py:91,183; source absent, algorithm per SURVEY.md §3.3/§8 a8):
//     cij[:,k] = np.correlate(x_i, x_j, 'full') / sqrt(sum x_i^2 * sum x_j^2)   (2W-1 lags)
//     cmax_k   = max_l cij[l,k]
//     tau_k    = (W - (argmax_l cij[l,k] + 1)) / fs          first maximum wins
// The kernel returns the integer lag W-1-argmax (tau = lag/fs, exact) and cmax_k; the
// median over pairs (MdCCM) is taken by the solve kernel.
//
// Cost: W^2 FP64 multiply-adds per (unit, pair) — this is the dominant kernel of the path
// and it is FP64-throughput bound (2*P*W^2 flop per unit against ~8*N*inc bytes), not
// HBM bound.
#include "nbls_internal.h"

namespace {

struct XArgs {
    const double* filt;       // [B][N][npts_pad]
    int64_t npts_pad;
    int nchans;
    int npairs;
    const int32_t* pair;      // [P][2]
    const int32_t* Wb;        // [B]
    const int32_t* incb;      // [B]
    const int32_t* unit_off;  // [B+1]
    const int32_t* unit_band; // [U]
    int vector_len;
    int32_t* lag;             // [B][VL][P]
    double* cmax;             // [B][VL][P]
};

__device__ inline bool better(double v1, int k1, double v2, int k2) {
    return (v1 > v2) || (v1 == v2 && k1 < k2);
}

// ------------------------------------------------------------------------------------
// v1: plain VALU kernel.  One workgroup per (unit, pair); both channel windows staged in
// LDS; thread t owns lags t, t+256, ...; consecutive lanes read consecutive LDS words
// (conflict-free), the partner sample is an LDS broadcast.  Two LDS reads per FMA make it
// LDS-issue bound — it is the always-correct general-purpose path and the checker for the
// MFMA kernel.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xcorr_simple_kernel(XArgs a) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x;
    const int64_t bid = blockIdx.x;
    const int u = (int)(bid / a.npairs);
    const int k = (int)(bid % a.npairs);
    const int band = a.unit_band[u];
    const int w = u - a.unit_off[band];
    const int W = a.Wb[band];
    const int64_t t0 = (int64_t)w * a.incb[band];
    const int ci = a.pair[2 * k], cj = a.pair[2 * k + 1];
    const double* xa = a.filt + ((int64_t)band * a.nchans + ci) * a.npts_pad + t0;
    const double* xb = a.filt + ((int64_t)band * a.nchans + cj) * a.npts_pad + t0;
    double* sa = sm;
    double* sb = sm + W;
    __shared__ double red_v[8];
    __shared__ int red_k[4];

    double qa = 0.0, qb = 0.0;
    for (int n = tid; n < W; n += 256) {
        const double va = xa[n], vb = xb[n];
        sa[n] = va;
        sb[n] = vb;
        qa += va * va;
        qb += vb * vb;
    }
    for (int off = 32; off > 0; off >>= 1) {
        qa += __shfl_down(qa, off, 64);
        qb += __shfl_down(qb, off, 64);
    }
    if ((tid & 63) == 0) { red_v[(tid >> 6) * 2] = qa; red_v[(tid >> 6) * 2 + 1] = qb; }
    __syncthreads();
    const double ssa = (red_v[0] + red_v[2]) + (red_v[4] + red_v[6]);
    const double ssb = (red_v[1] + red_v[3]) + (red_v[5] + red_v[7]);
    __syncthreads();

    double best = -__builtin_inf();
    int bestk = 0x7fffffff;
    const int nl = 2 * W - 1;
    for (int kk = tid; kk < nl; kk += 256) {
        const int d = kk - (W - 1);
        const int nlo = d < 0 ? -d : 0;
        const int nhi = d < 0 ? W : W - d;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
        int n = nlo;
        for (; n + 3 < nhi; n += 4) {
            c0 = __builtin_fma(sa[n + d], sb[n], c0);
            c1 = __builtin_fma(sa[n + d + 1], sb[n + 1], c1);
            c2 = __builtin_fma(sa[n + d + 2], sb[n + 2], c2);
            c3 = __builtin_fma(sa[n + d + 3], sb[n + 3], c3);
        }
        for (; n < nhi; ++n) c0 = __builtin_fma(sa[n + d], sb[n], c0);
        const double c = (c0 + c1) + (c2 + c3);
        if (better(c, kk, best, bestk)) { best = c; bestk = kk; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const int ok = __shfl_down(bestk, off, 64);
        if (better(ov, ok, best, bestk)) { best = ov; bestk = ok; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = best; red_k[tid >> 6] = bestk; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 4; ++i)
            if (better(red_v[i], red_k[i], best, bestk)) { best = red_v[i]; bestk = red_k[i]; }
        const int64_t o = ((int64_t)band * a.vector_len + w) * a.npairs + k;
        a.lag[o] = (W - 1) - bestk;
        a.cmax[o] = best / sqrt(ssa * ssb);
    }
}

__global__ void probe_mfma_f64_kernel(const double* a, const double* b, double* out) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[lane], b[lane], acc, 0, 0, 0);
    out[lane * 4 + 0] = acc[0];
    out[lane * 4 + 1] = acc[1];
    out[lane * 4 + 2] = acc[2];
    out[lane * 4 + 3] = acc[3];
}

}  // namespace

hipError_t nbls_launch_xcorr(nbls_handle* h) {
    XArgs a;
    a.filt = h->d_filt;
    a.npts_pad = h->npts_pad;
    a.nchans = h->nchans;
    a.npairs = h->npairs;
    a.pair = h->d_pair;
    a.Wb = h->d_W;
    a.incb = h->d_inc;
    a.unit_off = h->d_unit_off;
    a.unit_band = h->d_unit_band;
    a.vector_len = h->vector_len;
    a.lag = h->d_lag;
    a.cmax = h->d_cmax;
    if (h->nunits == 0) return hipSuccess;
    const int64_t nblocks = h->nunits * h->npairs;
    const size_t shm = (size_t)2 * h->maxW * sizeof(double);
    hipLaunchKernelGGL(xcorr_simple_kernel, dim3((unsigned)nblocks), dim3(256), shm, h->stream, a);
    h->tim.xcorr_launches = 1;
    return hipGetLastError();
}

hipError_t nbls_launch_probe_mfma(nbls_handle* h, const double* da, const double* db, double* dout) {
    hipLaunchKernelGGL(probe_mfma_f64_kernel, dim3(1), dim3(64), 0, h->stream, da, db, dout);
    return hipGetLastError();
}
