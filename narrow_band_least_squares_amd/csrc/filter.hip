// Per-band SOS band-pass (+ whole-trace taper) of the multichannel trace, for all
// (band, channel) series in one batch.
//
// Replaces helpers.py:124-139 of the reference (filter_data: st.copy(); obspy zero-phase
// Butterworth band-pass or scipy.signal.sosfilt causal Chebyshev-I; stf.taper(0.01)).
//
// An IIR cascade is a recurrence in time, and there are only B*N independent series
// (e.g. 384), far too few for 256 CUs.  So time is cut into chunks of NBLS_FILTER_CHUNK
// samples, one chunk per lane, and the recurrence is carried across chunks by the affine
// map of the 2S-dimensional DF2T state:
//
//   phase 1  every chunk is filtered from a ZERO initial state; only its end state e_c is kept
//   phase 2  per series, sequentially over chunks:  init_c = s;  s = M s + e_c   (M = A^C,
//            the zero-input transition of one chunk, computed on the host)
//   phase 3  every chunk is filtered again from init_c and the output is written
//
// Inside a chunk the arithmetic is scipy's DF2T recurrence, un-fused (this file is built
// with -ffp-contract=off), so the output differs from scipy.signal.sosfilt only by the
// rounding of the carried states.  Zero-phase = the same three phases on the time-reversed
// pass-1 output, in place.  The taper is multiplied in when the last pass writes.
//
// Memory access: lane <-> chunk means a lane's samples are 4 KiB apart, so each wave stages
// a 64-chunk x 32-sample tile through LDS (row stride 33 doubles: conflict-free row reads):
// global loads/stores move 256-B contiguous row segments, HBM sees every sample once per
// phase.  HBM-bound; algorithmic traffic = (2 reads + 1 write) x 8 B per sample per pass.
#include "nbls_internal.h"

namespace {

constexpr int C = NBLS_FILTER_CHUNK;
constexpr int T = NBLS_FILTER_TILE;

struct FilterArgs {
    const double* in;
    int64_t in_stride;
    int in_mod;
    double* out;
    int64_t out_stride;
    const double* sos;     // [B][S][6]
    double* cstate;        // [nchunks][nseries][2S]
    int nchans;
    int nseries;
    int64_t npts;
    int64_t nchunks;
    int reverse;
    int final_pass;
    const double* tl;
    const double* tr;
    int taper_len;
};

template <int S, int PHASE>
__global__ __launch_bounds__(64) void filter_chunk_kernel(FilterArgs a) {
    __shared__ double tile[64][T + 1];
    const int lane = threadIdx.x;
    const int q = blockIdx.y;
    const int band = q / a.nchans;
    const int64_t chunk0 = (int64_t)blockIdx.x * 64;
    const int64_t chunk = chunk0 + lane;
    const double* in = a.in + (int64_t)(q % a.in_mod) * a.in_stride;
    double* out = a.out + (int64_t)q * a.out_stride;

    double b0[S], b1[S], b2[S], a1[S], a2[S], s1[S], s2[S];
    const double* sos = a.sos + (int64_t)band * S * 6;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        b0[s] = sos[s * 6 + 0];
        b1[s] = sos[s * 6 + 1];
        b2[s] = sos[s * 6 + 2];
        a1[s] = sos[s * 6 + 4];
        a2[s] = sos[s * 6 + 5];
        s1[s] = 0.0;
        s2[s] = 0.0;
    }
    if (PHASE == 3 && chunk < a.nchunks) {
        const double* st = a.cstate + ((int64_t)chunk * a.nseries + q) * (2 * S);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            s1[s] = st[2 * s];
            s2[s] = st[2 * s + 1];
        }
    }

    for (int ti = 0; ti < C / T; ++ti) {
        if (chunk0 * C + (int64_t)ti * T >= a.npts) break;   // wave-uniform
        // cooperative load: 2 rows x 32 samples per instruction, 256-B contiguous segments
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int row = 2 * i + (lane >> 5);
            const int col = lane & 31;
            const int64_t p = (chunk0 + row) * C + (int64_t)ti * T + col;
            double v = 0.0;
            if (p < a.npts) {
                const int64_t g = a.reverse ? (a.npts - 1 - p) : p;
                v = in[g];
            }
            tile[row][col] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < T; ++t) {
            double v = tile[lane][t];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const double y = b0[s] * v + s1[s];
                s1[s] = (b1[s] * v - a1[s] * y) + s2[s];
                s2[s] = b2[s] * v - a2[s] * y;
                v = y;
            }
            if (PHASE == 3) tile[lane][t] = v;
        }
        __syncthreads();
        if (PHASE == 3) {
#pragma unroll 4
            for (int i = 0; i < 32; ++i) {
                const int row = 2 * i + (lane >> 5);
                const int col = lane & 31;
                const int64_t p = (chunk0 + row) * C + (int64_t)ti * T + col;
                if (p < a.npts) {
                    const int64_t g = a.reverse ? (a.npts - 1 - p) : p;
                    double v = tile[row][col];
                    if (a.final_pass) {
                        if (g < a.taper_len) v = v * a.tl[g];
                        else if (g >= a.npts - a.taper_len) v = v * a.tr[g - (a.npts - a.taper_len)];
                    }
                    out[g] = v;
                }
            }
            __syncthreads();
        }
    }
    if (PHASE == 1 && chunk < a.nchunks) {
        double* st = a.cstate + ((int64_t)chunk * a.nseries + q) * (2 * S);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            st[2 * s] = s1[s];
            st[2 * s + 1] = s2[s];
        }
    }
}

// phase 2: carry the state across chunks, one lane per series.
template <int S>
__global__ void filter_carry_kernel(double* cstate, const double* M, int nseries, int nchans,
                                    int64_t nchunks) {
    constexpr int D = 2 * S;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nseries) return;
    const int band = q / nchans;
    double m[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) m[i][j] = M[((int64_t)band * D + i) * D + j];
    double s[D], e[D], en[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { s[i] = 0.0; e[i] = cstate[(int64_t)q * D + i]; }
    for (int64_t c = 0; c < nchunks; ++c) {
        double* cur = cstate + ((int64_t)c * nseries + q) * D;
        if (c + 1 < nchunks) {
            const double* nx = cstate + ((int64_t)(c + 1) * nseries + q) * D;
#pragma unroll
            for (int i = 0; i < D; ++i) en[i] = nx[i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) cur[i] = s[i];
        double sn[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double acc = e[i];
#pragma unroll
            for (int j = 0; j < D; ++j) acc += m[i][j] * s[j];
            sn[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) { s[i] = sn[i]; e[i] = en[i]; }
    }
}

template <int S>
hipError_t run_pass(nbls_handle* h, const FilterArgs& base) {
    dim3 grid((unsigned)((h->nchunks + 63) / 64), (unsigned)base.nseries);
    hipLaunchKernelGGL((filter_chunk_kernel<S, 1>), grid, dim3(64), 0, h->stream, base);
    hipLaunchKernelGGL((filter_carry_kernel<S>), dim3((base.nseries + 63) / 64), dim3(64), 0,
                       h->stream, base.cstate, h->d_M, base.nseries, base.nchans, h->nchunks);
    hipLaunchKernelGGL((filter_chunk_kernel<S, 3>), grid, dim3(64), 0, h->stream, base);
    return hipGetLastError();
}

template <int S>
hipError_t run_filter(nbls_handle* h) {
    FilterArgs a;
    a.nchans = h->nchans;
    a.nseries = h->nbands * h->nchans;
    a.npts = h->npts;
    a.nchunks = h->nchunks;
    a.sos = h->d_sos;
    a.cstate = h->d_cstate;
    a.tl = h->d_tl;
    a.tr = h->d_tr;
    a.taper_len = h->taper_len;
    a.out = h->d_filt;
    a.out_stride = h->npts_pad;
    // pass 1: raw trace -> filtered buffer, forward in time
    a.in = h->d_trace;
    a.in_stride = h->npts_pad;
    a.in_mod = h->nchans;
    a.reverse = 0;
    a.final_pass = h->zero_phase ? 0 : 1;
    hipError_t e = run_pass<S>(h, a);
    if (e != hipSuccess || !h->zero_phase) return e;
    // pass 2: in place, backward in time
    a.in = h->d_filt;
    a.in_mod = a.nseries;
    a.reverse = 1;
    a.final_pass = 1;
    return run_pass<S>(h, a);
}

}  // namespace

// nsections == 0: the trace is already filtered; copy it (and apply the taper, if any).
__global__ void copy_taper_kernel(const double* in, double* out, int64_t stride, int64_t npts, int nchans,
                                  const double* tl, const double* tr, int taper_len) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ch = blockIdx.y;
    if (g >= npts || ch >= nchans) return;
    double v = in[ch * stride + g];
    if (g < taper_len) v = v * tl[g];
    else if (g >= npts - taper_len) v = v * tr[g - (npts - taper_len)];
    out[ch * stride + g] = v;
}

hipError_t nbls_launch_filter(nbls_handle* h) {
    if (h->nsections == 0) {
        hipLaunchKernelGGL(copy_taper_kernel, dim3((unsigned)((h->npts + 255) / 256), h->nchans), dim3(256), 0,
                           h->stream, h->d_trace, h->d_filt, h->npts_pad, h->npts, h->nchans, h->d_tl, h->d_tr,
                           h->taper_len);
        return hipGetLastError();
    }
    switch (h->nsections) {
        case 1: return run_filter<1>(h);
        case 2: return run_filter<2>(h);
        case 3: return run_filter<3>(h);
        case 4: return run_filter<4>(h);
        case 5: return run_filter<5>(h);
        case 6: return run_filter<6>(h);
        case 7: return run_filter<7>(h);
        case 8: return run_filter<8>(h);
        default: return hipErrorInvalidValue;
    }
}
