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
#include "wave_ops.h"

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
    const int32_t* win_off;   // [B] first window of each band's range (window sharding)
    const int32_t* unit_band; // [U]
    int vector_len;
    int32_t* lag;             // [B][VL][P]
    double* cmax;             // [B][VL][P]
    // f64-MFMA kernel only
    int S;                    // lag blocks (of 16) per tile = floor(16 / (N-1))
    int CS;                   // LDS elements per channel buffer, == 2 (mod 32)
    int PF;                   // zero padding in front of each channel window
    int u0;                   // first unit of this launch (window groups of a plan: nbls_launch_xcorr)
    int from_global;          // xcorr_simple_kernel: the two windows do not fit a CU's LDS (> 10 000 samples) and are read
                              // from global memory (L2) instead — no window length is refused
};

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
    const int u = a.u0 + (int)(bid / a.npairs);
    const int k = (int)(bid % a.npairs);
    const int band = a.unit_band[u];
    const int w = u - a.unit_off[band] + a.win_off[band];   // window index inside the band (global)
    const int W = a.Wb[band];
    const int64_t t0 = (int64_t)w * a.incb[band];
    const int ci = a.pair[2 * k], cj = a.pair[2 * k + 1];
    const double* xa = a.filt + ((int64_t)band * a.nchans + ci) * a.npts_pad + t0;
    const double* xb = a.filt + ((int64_t)band * a.nchans + cj) * a.npts_pad + t0;
    const double* sa = a.from_global ? xa : sm;
    const double* sb = a.from_global ? xb : sm + W;
    __shared__ double red_v[8];
    __shared__ int red_k[4];

    double qa = 0.0, qb = 0.0;
    for (int n = tid; n < W; n += 256) {
        const double va = xa[n], vb = xb[n];
        if (!a.from_global) { sm[n] = va; sm[W + n] = vb; }
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
    const double nrm_ab = ssa * ssb;                     // (square of the norm) maxima are compared by quotient, like np.argmax(cij / norm): better_q
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
        if (nbls_wave::better_q(c, kk, best, bestk, nrm_ab)) { best = c; bestk = kk; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const int ok = __shfl_down(bestk, off, 64);
        if (nbls_wave::better_q(ov, ok, best, bestk, nrm_ab)) { best = ov; bestk = ok; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = best; red_k[tid >> 6] = bestk; }
    __syncthreads();
    if (tid < 64) {
        for (int i = 1; i < 4; ++i)
            if (nbls_wave::better_q(red_v[i], red_k[i], best, bestk, nrm_ab)) { best = red_v[i]; bestk = red_k[i]; }
        if (!nbls_wave::finite_f64(ssa) || !nbls_wave::finite_f64(ssb)) {      // NaN / Inf samples: NumPy's semantics
            bestk = nbls_wave::nonfinite_argmax(sa, sb, W, tid);
            best = __builtin_nan("");
        }
        if (tid == 0) {
            const int64_t o = ((int64_t)band * a.vector_len + w) * a.npairs + k;
            a.lag[o] = (W - 1) - bestk;
            a.cmax[o] = best / sqrt(ssa * ssb);
        }
    }
}


// ------------------------------------------------------------------------------------
// v2: FP64 matrix-core kernel (v_mfma_f64_16x16x4_f64).  One workgroup per unit, one wave per
// "sliding" channel i.  The whole N-channel window sits in LDS once (zero padded, one buffer per
// channel).  For wave i the positive-lag correlations against ALL other channels are a
// Toeplitz product that maps onto 16x16 MFMA tiles with no wasted lanes:
//
//     C[r][c] = sum_n'  A[r][n'] * B[n'][c],   A[r][n'] = x_i[n' + r + D0]
//                                              B[n'][c] = x_j(c)[n' - 16 s(c)]
//     => C[r][c] = sum_n x_i[n + d] x_j[n],    d = D0 + 16 s(c) + r
//
// rows r = 16 consecutive lags, columns c = (partner j, lag block s): (N-1)*S of the 16 columns
// are live (14/16 for N = 8).  A tile step covers 16*S lags of every partner; its K loop runs
// only over the n' that can overlap (W - D0), so the triangular lag/overlap structure costs
// ~16*S/2 wasted samples per lag instead of W/2.  Negative lags of pair (i, j) are the positive
// lags of wave j against partner i, so every ordered (i, j) is one column family and the total
// is P * W^2 multiply-adds, as in the direct form.
//
// LDS reads per MFMA: two ds_read_b64 (one A, one B value per lane).  A lanes read <= 17
// consecutive doubles (broadcast + conflict free); B lanes read one double per (column, k):
// with the channel stride CS == 2 (mod 32) doubles and the 16-sample block shift, the 32
// addresses of a half-wave fall on 32 distinct bank pairs.
//
// Per-lane running (max, first index) over its 4 accumulator rows and all tile steps, then a
// shuffle reduce over the 4 lanes of a column, then one LDS hand-off to combine the two
// orderings of each pair.  Ties resolve to the smaller np.correlate index, like np.argmax.
// ------------------------------------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void xcorr_mfma_kernel(XArgs a) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;                    // sliding channel of this wave
    const int N = a.nchans;
    const int u = a.u0 + blockIdx.x;
    const int band = a.unit_band[u];
    const int w = u - a.unit_off[band] + a.win_off[band];   // window index inside the band (global)
    const int W = a.Wb[band];
    const int64_t t0 = (int64_t)w * a.incb[band];
    const int S = a.S, CS = a.CS, PF = a.PF;
    double* nrm = sm + (size_t)N * CS;          // [N] sum of squares
    double* cbv = nrm + N;                      // [N][16] best value per (wave, column)
    int* cbk = (int*)(cbv + N * 16);            // [N][16] its np.correlate index

    {   // stage: wave wv loads channel wv (zero padded) and its sum of squares
        double* ch = sm + (size_t)wv * CS;
        const double* src = a.filt + ((int64_t)band * N + wv) * a.npts_pad + t0;
        double q = 0.0;
        for (int n = lane; n < CS; n += 64) {
            const int idx = n - PF;
            const double v = (idx >= 0 && idx < W) ? src[idx] : 0.0;
            ch[n] = v;
            q += v * v;
        }
        for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
        if (lane == 0) nrm[wv] = q;
    }
    __syncthreads();

    const int c = lane & 15, kq = lane >> 4;
    const int ncol = (N - 1) * S;
    const bool colvalid = c < ncol;
    const int jj = colvalid ? c % (N - 1) : 0;
    const int s = colvalid ? c / (N - 1) : 0;
    const int j = jj + (jj >= wv ? 1 : 0);      // partner channel of this column
    const double* pa = sm + (size_t)wv * CS + PF + kq + c;        // + D0 + n'   (row r = lane & 15)
    const double* pb = sm + (size_t)j * CS + PF + kq - 16 * s;    // + n'
    double bestv = -__builtin_inf();
    int bestk = 0x7fffffff;
    const double nrm_wj = wv < j ? nrm[wv] * nrm[j] : nrm[j] * nrm[wv];    // (square of the norm; product in pair order ci < cj, as in the final division)
    const int step = 16 * S;
    for (int D0 = 0; D0 < W; D0 += step) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const int klen = W - D0;
        const double* qa = pa + D0;
        const double* qb = pb;
        int n0 = 0;
        for (; n0 + 28 < klen; n0 += 32) {
#pragma unroll
            for (int uu = 0; uu < 8; ++uu)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[n0 + 4 * uu], qb[n0 + 4 * uu], acc, 0, 0, 0);
        }
        for (; n0 < klen; n0 += 4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[n0], qb[n0], acc, 0, 0, 0);
        if (colvalid) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int d = D0 + 16 * s + kq + 4 * reg;
                if (d < W) {
                    const int kk = (wv < j) ? (W - 1 + d) : (W - 1 - d);
                    if (nbls_wave::better_q(acc[reg], kk, bestv, bestk, nrm_wj)) { bestv = acc[reg]; bestk = kk; }
                }
            }
        }
    }
    for (int off = 16; off <= 32; off <<= 1) {
        const double ov = __shfl_xor(bestv, off, 64);
        const int ok = __shfl_xor(bestk, off, 64);
        if (nbls_wave::better_q(ov, ok, bestv, bestk, nrm_wj)) { bestv = ov; bestk = ok; }
    }
    if (lane < 16) { cbv[wv * 16 + lane] = bestv; cbk[wv * 16 + lane] = bestk; }
    __syncthreads();
    if (tid < a.npairs) {
        const int ci = a.pair[2 * tid], cj = a.pair[2 * tid + 1];     // ci < cj
        double bv = -__builtin_inf();
        int bk = 0x7fffffff;
        const double nrm_p2 = nrm[ci] * nrm[cj];
        for (int ss = 0; ss < S; ++ss) {
            const int c1 = (cj - 1) + (N - 1) * ss;     // wave ci, partner cj
            if (nbls_wave::better_q(cbv[ci * 16 + c1], cbk[ci * 16 + c1], bv, bk, nrm_p2)) { bv = cbv[ci * 16 + c1]; bk = cbk[ci * 16 + c1]; }
            const int c2 = ci + (N - 1) * ss;           // wave cj, partner ci
            if (nbls_wave::better_q(cbv[cj * 16 + c2], cbk[cj * 16 + c2], bv, bk, nrm_p2)) { bv = cbv[cj * 16 + c2]; bk = cbk[cj * 16 + c2]; }
        }
        if (!nbls_wave::finite_f64(nrm[ci]) || !nbls_wave::finite_f64(nrm[cj])) {
            // NaN / Inf samples: NumPy's semantics (see wave_ops.h nonfinite_argmax), scanned by this one thread
            const double* xa = sm + (size_t)ci * CS + PF;
            const double* xb = sm + (size_t)cj * CS + PF;
            bool has_nan = false;
            int first_a = 0x7fffffff, last_b = -1;
            for (int n = 0; n < W; ++n) {
                const double va = xa[n], vb = xb[n];
                has_nan = has_nan || (va != va) || (vb != vb);
                if (fabs(va) == __builtin_inf() && n < first_a) first_a = n;
                if (fabs(vb) == __builtin_inf()) last_b = n;
            }
            int kk = 0x7fffffff;
            if (first_a != 0x7fffffff) kk = first_a;
            if (last_b >= 0 && W - 1 - last_b < kk) kk = W - 1 - last_b;
            bk = (has_nan || kk == 0x7fffffff) ? 0 : kk;
            bv = __builtin_nan("");
        }
        const int64_t o = ((int64_t)band * a.vector_len + w) * a.npairs + tid;
        a.lag[o] = (W - 1) - bk;
        a.cmax[o] = bv / sqrt(nrm_p2);
    }
}

__global__ void probe_mfma_f64_kernel(const double* a, const double* b, double* out) {
    const int lane = threadIdx.x;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[lane], b[lane], acc, 0, 0, 0);
    out[lane * 4 + 0] = acc[0];
    out[lane * 4 + 1] = acc[1];
    out[lane * 4 + 2] = acc[2];
    out[lane * 4 + 3] = acc[3];
}

}  // namespace

// The general correlators for the units [ub, ue) (windows of up to gW samples).  impl: 1 plain VALU, 2 f64 MFMA,
// 0 the faster one that applies.
static hipError_t launch_general_range(nbls_handle* h, int64_t ub, int64_t ue, int gW, int impl, int* used) {
    XArgs a{};
    a.filt = h->d_filt;
    a.npts_pad = h->npts_pad;
    a.nchans = h->nchans;
    a.npairs = h->npairs;
    a.pair = h->d_pair;
    a.Wb = h->d_W;
    a.incb = h->d_inc;
    a.unit_off = h->d_unit_off;
    a.win_off = h->d_win_off;
    a.unit_band = h->d_unit_band;
    a.vector_len = h->vector_len;
    a.lag = h->d_lag;
    a.cmax = h->d_cmax;
    a.u0 = (int)ub;
    const int64_t nu = ue - ub;
    if (nu <= 0) return hipSuccess;
    // f64-MFMA kernel: needs one wave per channel (N <= 16) and the N-channel window in LDS
    const int N = h->nchans;
    bool mfma_ok = N >= 3 && N <= 16 && h->npairs <= 64 * N;
    size_t shm_m = 0;
    if (mfma_ok) {
        a.S = 16 / (N - 1);
        a.PF = 16 * (a.S - 1);
        int cs = a.PF + gW + 32;
        cs += ((2 - cs) % 32 + 32) % 32;            // CS == 2 (mod 32)
        a.CS = cs;
        shm_m = ((size_t)N * cs + N + N * 16) * sizeof(double) + (size_t)N * 16 * sizeof(int);
        if (shm_m > 160 * 1024) mfma_ok = false;
    }
    if (impl == 2 && !mfma_ok) return hipErrorInvalidValue;
    if (mfma_ok && impl != 1) {
        hipError_t e = hipFuncSetAttribute((const void*)xcorr_mfma_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_m);
        if (e != hipSuccess) return e;
        *used = 2;
        hipLaunchKernelGGL(xcorr_mfma_kernel, dim3((unsigned)nu), dim3(64 * N), shm_m, h->stream, a);
        return hipGetLastError();
    }
    const int64_t nblocks = nu * h->npairs;
    a.from_global = (size_t)2 * gW * sizeof(double) > 160 * 1024 ? 1 : 0;
    const size_t shm = a.from_global ? 0 : (size_t)2 * gW * sizeof(double);
    if (shm > 48 * 1024) {      // long windows (W up to 10000): opt in to more than the default dynamic LDS
        hipError_t e = hipFuncSetAttribute((const void*)xcorr_simple_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
    }
    *used = 1;
    hipLaunchKernelGGL(xcorr_simple_kernel, dim3((unsigned)nblocks), dim3(256), shm, h->stream, a);
    return hipGetLastError();
}

// The correlation stage of a pass.  The plan's bands come in window groups (h->wgroups: consecutive bands of one
// window length): each group takes the int8 screening path when its channel images fit a CU's LDS (possibly with
// partner groups) and a general correlator otherwise — a long-window band of an adaptive-window plan no longer sends
// every band of the plan to the 50x slower kernel.
hipError_t nbls_launch_xcorr(nbls_handle* h) {
    if (h->nunits == 0) return hipSuccess;
    h->tim.xcorr_launches = 0;
    h->tim.xcorr_fallback_bands = 0;
    h->bev_used = 0;
    int64_t launches = 0;
    int used = 0, fallback_bands = 0;
    bool any_screen = false;
    for (const nbls_wgroup& g : h->wgroups) {
        if (g.u1 <= g.u0) continue;
        hipError_t e;
        if (h->xcorr_impl == 3 && g.screen) {
            any_screen = true;
            e = nbls_launch_xcorr_screen_range(h, g.u0, g.u1, g.W, &launches);
        } else {
            int u_ = 0;
            e = launch_general_range(h, g.u0, g.u1, g.W, h->xcorr_impl == 3 ? 0 : h->xcorr_impl, &u_);
            used = u_ > used ? u_ : used;
            fallback_bands += g.b1 - g.b0;
            if (h->xcorr_impl != 3) ++launches;
            // per-batch solves (nbls_execute_stages: fuse_solve): a window group that does not take the screening path
            // is one batch of its own — without this its units were never solved (nbls_xcorr_screen_finish marks the
            // solve stage done for the whole pass)
            if (e == hipSuccess && h->fuse_solve) {
                e = nbls_launch_solve_range(h, g.u0, g.u1 - g.u0, h->stream);
                if (e == hipSuccess) e = nbls_queue_result_batch(h, g.u0, g.u1, h->stream);
            }
        }
        if (e != hipSuccess) return e;
    }
    if (h->xcorr_impl == 3 && (any_screen || h->fuse_solve)) {
        h->xcorr_impl_used = 3;
        h->tim.xcorr_fallback_bands = fallback_bands;          // bands of this pass that ran on a general correlator
        return nbls_xcorr_screen_finish(h, launches);
    }
    h->xcorr_impl_used = used;
    h->tim.xcorr_launches = launches;
    return hipGetLastError();
}

hipError_t nbls_launch_probe_mfma(nbls_handle* h, const double* da, const double* db, double* dout) {
    hipLaunchKernelGGL(probe_mfma_f64_kernel, dim3(1), dim3(64), 0, h->stream, da, db, dout);
    return hipGetLastError();
}
