// Per-band SOS band-pass (+ whole-trace taper) of the multichannel trace, for all
// (band, channel) series in one batch.
//
// Replaces helpers.py:124-139 of the reference (filter_data: st.copy(); obspy zero-phase
// Butterworth band-pass or scipy.signal.sosfilt causal Chebyshev-I; stf.taper(0.01)).
//
// An IIR cascade is a recurrence in time, and there are only B*N independent series
// (e.g. 384), far too few for 256 CUs.  Time is cut into chunks of C = NBLS_FILTER_CHUNK
// samples and the recurrence is carried across chunks through the affine map of the
// 2S-dimensional DF2T state  s_out = M s_in + e  (M = A^C, e = zero-state end state):
//
//   states   e_c of every chunk: the zero-state end state is LINEAR in the chunk's samples,
//            e_c = sum_t A^(C-1-t) g x_t, so it is a small GEMV with host-computed weights —
//            one wave per chunk, lanes along time, fully coalesced, no recurrence.
//   carry    hierarchical scan of s <- M s + e_c over the chunks of a series: groups of 64 chunks
//            are scanned locally in parallel, the few group totals sequentially (powers of M from
//            the host), the group-in state is folded into each chunk's start state by the next step.
//   apply    every chunk is filtered by the real recurrence from its start state and written
//            (one lane per chunk; 64-chunk x 32-sample tiles staged through LDS so that global
//            accesses are 256-B row segments; the next tile is prefetched into registers while the
//            current one is filtered).
//
// Inside a chunk the arithmetic is scipy's DF2T recurrence, un-fused (this file is built with
// -ffp-contract=off), so the output differs from scipy.signal.sosfilt only by the rounding of the
// carried start states.  Zero-phase = the same steps on the time-reversed pass-1 output, in
// place.  The taper is multiplied in when the last pass writes.  HBM-bound by construction:
// 2 reads + 1 write of 8 B per sample per pass.
#include "nbls_internal.h"
#include "wave_ops.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int C = NBLS_FILTER_CHUNK;
constexpr int T = NBLS_FILTER_TILE;
constexpr int G = NBLS_FILTER_GROUP;

struct FilterArgs {
    const double* in;
    int64_t in_stride;
    int in_mod;
    double* out;
    int64_t out_stride;
    const double* sos;     // [B][S][6]
    const double* fw;      // [B][C][2S]   state weights  w_t = A^(C-1-t) g
    const double* mpow;    // [B][G+1][2S][2S]  powers of M = A^C
    double* cstate;        // [nchunks][nseries][2S]   e_c, then local start states
    double* gend;          // [ngroups][nseries][2S]   group totals
    double* gin;           // [ngroups][nseries][2S]   state entering each group
    int nchans;
    int nseries;
    int ch0, nch;          // the channels [ch0, ch0 + nch) of every band are processed by this launch (nsub = nbands * nch series):
    int nsub;              // a pass may run channel by channel while the trace is still going up (nbls_execute, row events)
    int64_t npts;
    int64_t nchunks;
    int ngroups;
    int reverse;
    int64_t plen;          // length of the pass's index space: npts forward, nchunks*C backward (see run_filter)
    double* cstate_next;   // forward pass of a zero-phase filter: chunk states of the backward pass, else NULL
    int final_pass;
    const double* tl;
    const double* tr;
    int taper_len;
    double* tstate;        // [nseries][C/T][nchunks][2S] forward states at the tile boundaries (recompute != 0)
    int recompute;         // zero-phase without materialising the forward output: 1 = forward pass (stores the tile
                           // states, writes no samples), 2 = backward pass (rebuilds each tile's forward output from
                           // the raw trace and the stored state, then runs the backward recurrence over it)
    const double* init;    // [nseries][2S] state entering the first chunk (time-segmented filtering), else NULL = zero
    double* fin;           // [nseries][2S] state after the last (whole) chunk, else NULL
};

// series (band, channel) of the launch's ql-th series: all channels -> ql itself
__device__ __forceinline__ int series_of(const FilterArgs& a, int ql) {
    return a.nch == a.nchans ? ql : (ql / a.nch) * a.nchans + a.ch0 + ql % a.nch;
}

// ---- states: one wave per (series, chunk) ----
template <int S>
__global__ __launch_bounds__(256) void filter_state_kernel(FilterArgs a) {
    constexpr int D = 2 * S;
    const int lane = threadIdx.x & 63;
    const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int q = series_of(a, blockIdx.y);
    if (chunk >= a.nchunks) return;
    const int band = q / a.nchans;
    const double* in = a.in + (int64_t)(q % a.in_mod) * a.in_stride;
    const double* fw = a.fw + (int64_t)band * C * D;
    double acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.0;
    const int64_t p0 = chunk * C;
    if (p0 + C <= a.plen) {          // the end state of a partial last chunk is never used
#pragma unroll
        for (int k = 0; k < C / 64; ++k) {
            const int t = k * 64 + lane;
            const int64_t p = p0 + t;
            const int64_t g = a.reverse ? (a.plen - 1 - p) : p;
            const double x = g < a.npts ? in[g] : 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] += fw[t * D + d] * x;
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = nbls_wave::sum_f64(acc[d]);
    if (lane == 0) {
        double* st = a.cstate + (chunk * a.nseries + q) * D;
#pragma unroll
        for (int d = 0; d < D; ++d) st[d] = acc[d];
    }
}

// ---- states on the FP64 matrix cores (forward pass, D = 2, 4, 8 or 16) ----
// e[(band, d)][chunk] = sum_t w[(band, d)][t] x[chunk][t] is a GEMM with M = B*D rows (weights), N = all
// chunks of all channels, K = C samples.  A workgroup owns one tile of 16 rows (its weights are staged once
// in LDS, row stride C + 4 doubles: conflict-free fragment reads) and walks column tiles of 16 consecutive
// chunks of one channel with v_mfma_f64_16x16x4_f64 (A[i][k]: lane i + 16k, B[k][j]: lane j + 16k,
// D[i][j]: lane j + 16 (i % 4), register i / 4).  The trace is read straight from L2 (each 32-byte piece
// of a chunk row serves four K steps through L1); the weights are read once per workgroup.
typedef double d4f __attribute__((ext_vector_type(4)));

template <int S>
__global__ __launch_bounds__(256) void filter_state_mfma_kernel(FilterArgs a, int nbands) {
    constexpr int D = 2 * S;
    constexpr int P = C + 4;
    extern __shared__ double wts[];                 // [16][P]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row0 = blockIdx.x * 16;               // first (band, d) row of this workgroup
    for (int idx = tid; idx < 16 * C; idx += 256) {
        const int r = idx / C, t = idx % C;
        const int grow = row0 + r, band = grow / D, d = grow % D;
        wts[r * P + t] = band < nbands ? a.fw[((int64_t)band * C + t) * D + d] : 0.0;
    }
    __syncthreads();
    const int tpc = (int)((a.nchunks + 15) / 16);   // column tiles per channel
    const int ntile = a.nch * tpc;
    const int i = lane & 15, k = lane >> 4;
    // The sum over the samples may run in any order as long as both operands agree on it: K slot k of
    // step (m, u) stands for sample 16 m + 4 k + u, so that a lane fetches 32 contiguous bytes of its
    // chunk per four steps (whole 128-byte lines per wave) instead of 8 bytes per step.
    const double* wrow = wts + i * P + 4 * k;
    for (int ct = blockIdx.y * 4 + wv; ct < ntile; ct += gridDim.y * 4) {
        const int ch = a.ch0 + ct / tpc;
        const int64_t chunk = (int64_t)(ct % tpc) * 16 + i;          // this lane's column (B operand)
        const double* x = a.in + (int64_t)ch * a.in_stride + chunk * C + 4 * k;
        const int64_t left = chunk < a.nchunks ? a.npts - (chunk * C + 4 * k) : 0;   // valid samples from x[0]
        d4f acc = {0.0, 0.0, 0.0, 0.0};
        auto fetch = [&](int m, double (&v)[4]) {
            const int64_t off = 16 * m;
            if (off + 4 <= left) {
                const double2 p0 = *(const double2*)(x + off), p1 = *(const double2*)(x + off + 2);
                v[0] = p0.x; v[1] = p0.y; v[2] = p1.x; v[3] = p1.y;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = off + u < left ? x[off + u] : 0.0;
            }
        };
        double cur[4], nxt[4];
        fetch(0, cur);
        for (int m = 0; m < C / 16; ++m) {
            if (m + 1 < C / 16) fetch(m + 1, nxt);                  // in flight during this block's four products
            const double2 w0 = *(const double2*)(wrow + 16 * m), w1 = *(const double2*)(wrow + 16 * m + 2);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w0.x, cur[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w0.y, cur[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w1.x, cur[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w1.y, cur[3], acc, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int grow = row0 + 4 * r + k;       // D row = 4 * register + (lane >> 4)
            const int band = grow / D, d = grow % D;
            if (band < nbands && chunk < a.nchunks) {
                const int q = band * a.nchans + ch;
                a.cstate[(chunk * a.nseries + q) * D + d] = acc[r];
            }
        }
    }
}

// ---- carry, level 1: local scan inside each group of G chunks (lane per (group, series)) ----
template <int S>
__global__ void filter_carry_local_kernel(FilterArgs a) {
    constexpr int D = 2 * S;
    const int ql = blockIdx.x * blockDim.x + threadIdx.x;
    const int gi = blockIdx.y;
    if (ql >= a.nsub) return;
    const int q = series_of(a, ql);
    const int band = q / a.nchans;
    const double* M = a.mpow + ((int64_t)band * (G + 1) + 1) * D * D;      // M^1
    double m[D][D], s[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        s[i] = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) m[i][j] = M[i * D + j];
    }
    const int64_t c0 = (int64_t)gi * G;
    const int64_t c1 = (c0 + G < a.nchunks) ? c0 + G : a.nchunks;
    for (int64_t c = c0; c < c1; ++c) {
        double* cur = a.cstate + (c * a.nseries + q) * D;
        double e[D], sn[D];
#pragma unroll
        for (int i = 0; i < D; ++i) e[i] = cur[i];
#pragma unroll
        for (int i = 0; i < D; ++i) cur[i] = s[i];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double acc = e[i];
#pragma unroll
            for (int j = 0; j < D; ++j) acc += m[i][j] * s[j];
            sn[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) s[i] = sn[i];
    }
    double* ge = a.gend + ((int64_t)gi * a.nseries + q) * D;
#pragma unroll
    for (int i = 0; i < D; ++i) ge[i] = s[i];
}

// ---- carry, level 2: sequential over the groups (lane per series) ----
template <int S>
__global__ void filter_carry_groups_kernel(FilterArgs a) {
    constexpr int D = 2 * S;
    const int ql = blockIdx.x * blockDim.x + threadIdx.x;
    if (ql >= a.nsub) return;
    const int q = series_of(a, ql);
    const int band = q / a.nchans;
    double s[D];
#pragma unroll
    for (int i = 0; i < D; ++i) s[i] = a.init ? a.init[(int64_t)q * D + i] : 0.0;
    for (int gi = 0; gi < a.ngroups; ++gi) {
        double* gin = a.gin + ((int64_t)gi * a.nseries + q) * D;
        const double* ge = a.gend + ((int64_t)gi * a.nseries + q) * D;
#pragma unroll
        for (int i = 0; i < D; ++i) gin[i] = s[i];
        const int64_t c0 = (int64_t)gi * G;
        const int len = (int)((c0 + G < a.nchunks ? c0 + G : a.nchunks) - c0);
        const double* Mg = a.mpow + ((int64_t)band * (G + 1) + len) * D * D;     // M^len
        double sn[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double acc = ge[i];
#pragma unroll
            for (int j = 0; j < D; ++j) acc += Mg[i * D + j] * s[j];
            sn[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) s[i] = sn[i];
    }
    if (a.fin) {
#pragma unroll
        for (int i = 0; i < D; ++i) a.fin[(int64_t)q * D + i] = s[i];
    }
}

// ---- apply: recurrence from the chunk's start state, one lane per chunk ----
// MODE (= FilterArgs::recompute, fixed at compile time so that each form carries only its own instructions: the
// kernel is bound by instruction issue, ~50 vector instructions per sample): 0 plain pass, 1 forward pass of the
// recompute form (tile states out, no samples), 2 its backward pass (forward output rebuilt per tile).
template <int S, int MODE>
__global__ __launch_bounds__(64) void filter_apply_kernel(FilterArgs a) {
    constexpr int D = 2 * S;
    __shared__ double tile[64][T + 1];
    __shared__ double wtile[T * D];      // backward-pass state weights of the tile's samples (fused states)
    const int lane = threadIdx.x;
    const int q = series_of(a, blockIdx.y);
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
    if (chunk < a.nchunks) {
        // start state = local prefix + M^j (state entering the group), j = index inside the group
        const int gi = (int)(chunk / G), j = (int)(chunk % G);
        const double* loc = a.cstate + (chunk * a.nseries + q) * D;
        const double* gin = a.gin + ((int64_t)gi * a.nseries + q) * D;
        const double* Mj = a.mpow + ((int64_t)band * (G + 1) + j) * D * D;
        double st[D], gv[D];
#pragma unroll
        for (int i = 0; i < D; ++i) gv[i] = gin[i];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double acc = loc[i];
#pragma unroll
            for (int k = 0; k < D; ++k) acc += Mj[i * D + k] * gv[k];
            st[i] = acc;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) { s1[s] = st[2 * s]; s2[s] = st[2 * s + 1]; }
    }

    const double* fwb = a.fw + (int64_t)band * C * D;
    double e2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) e2[d] = 0.0;
    // tile (64 chunks x T samples): instruction i moves the 64/T rows from (64/T) i on (8T-byte contiguous segments)
    double pre[T];
    // all 64 chunks of this workgroup inside the trace (every workgroup but one): no per-sample bounds tests
    const bool full = (chunk0 + 64) * C <= a.plen && (!a.reverse || chunk0 * C >= a.plen - a.npts) &&
                      (a.reverse || (chunk0 + 64) * C <= a.npts);
    auto fetch = [&](int ti) {
        if (full) {
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const int row = (64 / T) * i + lane / T;
                const int64_t p = (chunk0 + row) * C + (int64_t)ti * T + lane % T;
                pre[i] = in[a.reverse ? (a.plen - 1 - p) : p];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int row = (64 / T) * i + lane / T;
            const int64_t p = (chunk0 + row) * C + (int64_t)ti * T + lane % T;
            const int64_t g = a.reverse ? (a.plen - 1 - p) : p;
            pre[i] = (p < a.plen && g < a.npts) ? in[g] : 0.0;
        }
    };
    fetch(0);
    for (int ti = 0; ti < C / T; ++ti) {
        if (chunk0 * C + (int64_t)ti * T >= a.plen) break;   // wave-uniform
#pragma unroll
        for (int i = 0; i < T; ++i) tile[(64 / T) * i + lane / T][lane % T] = pre[i];
        double f1[S], f2[S];                                   // forward state entering this tile (backward pass of the recompute form)
        if (MODE == 1 && chunk < a.nchunks) {
            double* ts = a.tstate + (((int64_t)q * (C / T) + ti) * a.nchunks + chunk) * D;
#pragma unroll
            for (int s = 0; s < S; ++s) { ts[2 * s] = s1[s]; ts[2 * s + 1] = s2[s]; }
        } else if (MODE == 2) {
            // backward chunk c covers forward chunk nchunks-1-c, backward tile ti its forward tile C/T-1-ti
            const int64_t fc = a.nchunks - 1 - (chunk < a.nchunks ? chunk : a.nchunks - 1);
            const double* ts = a.tstate + (((int64_t)q * (C / T) + (C / T - 1 - ti)) * a.nchunks + fc) * D;
#pragma unroll
            for (int s = 0; s < S; ++s) { f1[s] = ts[2 * s]; f2[s] = ts[2 * s + 1]; }
        }
        if (a.cstate_next)
            for (int idx = lane; idx < T * D; idx += 64) {
                const int tl_ = idx / D, d = idx % D;
                wtile[idx] = fwb[(size_t)(C - 1 - (ti * T + tl_)) * D + d];
            }
        __syncthreads();
        if (ti + 1 < C / T) fetch(ti + 1);                    // in flight while this tile is filtered
        if constexpr (MODE == 2) {
            // the tile holds RAW samples in backward order: the forward recurrence runs through it from its last
            // column to its first (ascending time) from the stored state — the same operations on the same values
            // as in the forward pass, so the samples are bit-identical to the ones that pass had in hand — and
            // leaves the forward output in place; beyond the end of the trace that output counts as zero
            auto rebuild = [&](auto masked) {
#pragma unroll 4
                for (int t = T - 1; t >= 0; --t) {
                    double v = tile[lane][t];
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const double y = b0[s] * v + f1[s];
                        f1[s] = (b1[s] * v - a1[s] * y) + f2[s];
                        f2[s] = b2[s] * v - a2[s] * y;
                        v = y;
                    }
                    if (decltype(masked)::value) {
                        const int64_t p = chunk * C + (int64_t)ti * T + t;
                        v = (p < a.plen && a.plen - 1 - p < a.npts) ? v : 0.0;
                    }
                    tile[lane][t] = v;
                }
            };
            if (full) rebuild(std::false_type{}); else rebuild(std::true_type{});
        }
        auto recur = [&](auto masked) {
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
                if (MODE != 1) tile[lane][t] = v;         // (the forward pass of the recompute form writes no samples)
                if (a.cstate_next) {
                    // zero-state end state of the BACKWARD pass's chunk that covers the same samples
                    // (reversed): weight index C-1-tt; samples beyond the trace count as zeros there.  Fused
                    // multiply-add: these sums are start states of the carry scan, not samples of the recurrence
                    double yv = v;
                    if (decltype(masked)::value) yv = (chunk * C + ti * T + t < a.npts) ? v : 0.0;
#pragma unroll
                    for (int d = 0; d < D; ++d) e2[d] = __builtin_fma(wtile[t * D + d], yv, e2[d]);
                }
            }
        };
        if (full) recur(std::false_type{}); else recur(std::true_type{});
        __syncthreads();
        if (MODE != 1)
#pragma unroll 4
        for (int i = 0; i < T; ++i) {
            const int row = (64 / T) * i + lane / T;
            const int col = lane % T;
            const int64_t p = (chunk0 + row) * C + (int64_t)ti * T + col;
            const int64_t g = a.reverse ? (a.plen - 1 - p) : p;
            if (full || (p < a.plen && g < a.npts)) {
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
    if (a.cstate_next && chunk < a.nchunks) {
        double* st = a.cstate_next + ((a.nchunks - 1 - chunk) * a.nseries + q) * D;
#pragma unroll
        for (int d = 0; d < D; ++d) st[d] = e2[d];
    }
}

template <int S>
hipError_t run_pass(nbls_handle* h, const FilterArgs& a, bool states_ready) {
    constexpr int D = 2 * S;
    const bool no_mfma = h->opt.filter_nomfma != 0;                        // option: VALU state kernel
    if (!states_ready && !a.reverse && (16 % D) == 0 && a.in_mod == a.nchans && !no_mfma) {
        const int nbands = a.nseries / a.nchans;
        const int rows = (nbands * D + 15) / 16;
        const size_t shm = (size_t)16 * (C + 4) * sizeof(double);
        hipError_t e = hipFuncSetAttribute((const void*)filter_state_mfma_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
        const int tiles = a.nch * (int)((a.nchunks + 15) / 16);
        // column tiles per wave chosen so that the grid is about one round of two workgroups per CU
        int per_wave = (rows * tiles + 4 * 512 - 1) / (4 * 512);
        if (per_wave < 1) per_wave = 1;
        int splits = (tiles + 4 * per_wave - 1) / (4 * per_wave);
        if (splits < 1) splits = 1;
        hipLaunchKernelGGL((filter_state_mfma_kernel<S>), dim3((unsigned)rows, (unsigned)splits), dim3(256), shm, h->stream, a, nbands);
    } else if (!states_ready)
        hipLaunchKernelGGL((filter_state_kernel<S>), dim3((unsigned)((h->nchunks + 3) / 4), (unsigned)a.nsub),
                           dim3(256), 0, h->stream, a);
    hipLaunchKernelGGL((filter_carry_local_kernel<S>), dim3((a.nsub + 63) / 64, a.ngroups), dim3(64), 0,
                       h->stream, a);
    hipLaunchKernelGGL((filter_carry_groups_kernel<S>), dim3((a.nsub + 63) / 64), dim3(64), 0, h->stream, a);
    const dim3 agrid((unsigned)((h->nchunks + 63) / 64), (unsigned)a.nsub);
    if (a.recompute == 1) hipLaunchKernelGGL((filter_apply_kernel<S, 1>), agrid, dim3(64), 0, h->stream, a);
    else if (a.recompute == 2) hipLaunchKernelGGL((filter_apply_kernel<S, 2>), agrid, dim3(64), 0, h->stream, a);
    else hipLaunchKernelGGL((filter_apply_kernel<S, 0>), agrid, dim3(64), 0, h->stream, a);
    return hipGetLastError();
}

template <int S>
hipError_t run_filter(nbls_handle* h, int ch0, int nch) {
    FilterArgs a;
    a.nchans = h->nchans;
    a.nseries = h->nbands * h->nchans;
    a.ch0 = ch0;
    a.nch = nch;
    a.nsub = h->nbands * nch;
    a.npts = h->npts;
    a.nchunks = h->nchunks;
    a.ngroups = (int)((h->nchunks + G - 1) / G);
    a.sos = h->d_sos;
    a.fw = h->d_fw;
    a.mpow = h->d_M;
    a.cstate = h->d_cstate;
    a.gend = h->d_gend;
    a.gin = h->d_gin;
    a.tl = h->d_tl;
    a.tr = h->d_tr;
    a.taper_len = h->taper_len;
    a.init = nullptr;
    a.fin = nullptr;
    a.out = h->d_filt;
    a.out_stride = h->npts_pad;
    // pass 1: raw trace -> filtered buffer, forward in time
    a.in = h->d_trace;
    a.in_stride = h->npts_pad;
    a.in_mod = h->nchans;
    a.reverse = 0;
    a.plen = h->npts;
    a.final_pass = h->zero_phase ? 0 : 1;
    // zero-phase: the forward apply also accumulates the chunk states of the backward pass (it has
    // every output sample in hand), which saves the backward pass's read of the whole buffer
    const bool fuse = h->zero_phase && !h->opt.filter_nofuse;
    a.cstate_next = fuse ? h->d_cstate2 : nullptr;
    // zero-phase, recompute form: the forward output is never written.  The forward apply keeps the recurrence state
    // at every tile boundary (2S doubles per T samples: an eighth of the output at two sections) and the backward
    // apply rebuilds each tile's forward output from the raw trace (55 MB shared by all bands: it stays in the
    // caches) before it runs the backward recurrence over it: one 8-byte write per sample instead of two writes
    // and a read.
    const bool recompute = fuse && h->d_tstate;
    a.tstate = h->d_tstate;
    a.recompute = recompute ? 1 : 0;
    hipError_t e = run_pass<S>(h, a, false);
    if (e != hipSuccess || !h->zero_phase) return e;
    // pass 2: in place, backward in time.  The backward index space is padded at its START to a whole
    // number of chunks (zeros in, zero state: no effect on the output), so that its chunks coincide with
    // the forward chunks: backward chunk c covers forward chunk nchunks-1-c.
    a.in = h->d_filt;
    a.in_mod = a.nseries;
    if (recompute) { a.in = h->d_trace; a.in_mod = h->nchans; a.recompute = 2; }
    a.reverse = 1;
    a.plen = h->nchunks * C;
    a.final_pass = 1;
    a.cstate_next = nullptr;
    if (fuse) a.cstate = h->d_cstate2;
    return run_pass<S>(h, a, fuse);
}

// One causal pass over the resident segment, continuing from a given state (time-segmented filtering of traces whose
// filtered form does not fit HBM): reverse == 0: raw trace -> filtered buffer, forward in time;
// reverse == 1: the filtered buffer in place, backward in time (second half of a zero-phase filter).  No taper (the
// caller applies it at global positions).  The state leaving the last WHOLE chunk is written to d_fin, so every segment
// but the last (forward) / but the one processed first (backward) must be a whole number of chunks long.
template <int S>
hipError_t run_filter_segment(nbls_handle* h, int reverse, const double* d_init, double* d_fin) {
    FilterArgs a;
    a.nchans = h->nchans;
    a.nseries = h->nbands * h->nchans;
    a.ch0 = 0;
    a.nch = h->nchans;
    a.nsub = a.nseries;
    a.npts = h->npts;
    a.nchunks = h->nchunks;
    a.ngroups = (int)((h->nchunks + G - 1) / G);
    a.sos = h->d_sos;
    a.fw = h->d_fw;
    a.mpow = h->d_M;
    a.cstate = h->d_cstate;
    a.gend = h->d_gend;
    a.gin = h->d_gin;
    a.tl = h->d_tl;
    a.tr = h->d_tr;
    a.taper_len = 0;
    a.final_pass = 0;
    a.cstate_next = nullptr;
    a.tstate = nullptr;
    a.recompute = 0;
    a.init = d_init;
    a.fin = d_fin;
    a.out = h->d_filt;
    a.out_stride = h->npts_pad;
    if (!reverse) {
        a.in = h->d_trace;
        a.in_stride = h->npts_pad;
        a.in_mod = h->nchans;
        a.reverse = 0;
        a.plen = h->npts;
    } else {
        a.in = h->d_filt;
        a.in_stride = h->npts_pad;
        a.in_mod = a.nseries;
        a.reverse = 1;
        a.plen = h->nchunks * C;
    }
    return run_pass<S>(h, a, false);
}

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

}  // namespace

hipError_t nbls_launch_filter_segment(nbls_handle* h, int reverse, const double* d_init, double* d_fin) {
    switch (h->nsections) {
        case 1: return run_filter_segment<1>(h, reverse, d_init, d_fin);
        case 2: return run_filter_segment<2>(h, reverse, d_init, d_fin);
        case 3: return run_filter_segment<3>(h, reverse, d_init, d_fin);
        case 4: return run_filter_segment<4>(h, reverse, d_init, d_fin);
        case 5: return run_filter_segment<5>(h, reverse, d_init, d_fin);
        case 6: return run_filter_segment<6>(h, reverse, d_init, d_fin);
        case 7: return run_filter_segment<7>(h, reverse, d_init, d_fin);
        case 8: return run_filter_segment<8>(h, reverse, d_init, d_fin);
        default: return hipErrorInvalidValue;
    }
}

// The channels [ch0, ch0 + nch) of every band (all channels: the usual single launch; a subset: the rows of a trace
// that is still going up, filtered as they land — identical results, every (band, channel) series is independent).
hipError_t nbls_launch_filter(nbls_handle* h, int ch0, int nch) {
    if (ch0 < 0 || nch < 1 || ch0 + nch > h->nchans) return hipErrorInvalidValue;
    if (h->nsections == 0) {
        hipLaunchKernelGGL(copy_taper_kernel, dim3((unsigned)((h->npts + 255) / 256), nch), dim3(256), 0,
                           h->stream, h->d_trace + (size_t)ch0 * h->npts_pad, h->d_filt + (size_t)ch0 * h->npts_pad, h->npts_pad,
                           h->npts, nch, h->d_tl, h->d_tr, h->taper_len);
        return hipGetLastError();
    }
    switch (h->nsections) {
        case 1: return run_filter<1>(h, ch0, nch);
        case 2: return run_filter<2>(h, ch0, nch);
        case 3: return run_filter<3>(h, ch0, nch);
        case 4: return run_filter<4>(h, ch0, nch);
        case 5: return run_filter<5>(h, ch0, nch);
        case 6: return run_filter<6>(h, ch0, nch);
        case 7: return run_filter<7>(h, ch0, nch);
        case 8: return run_filter<8>(h, ch0, nch);
        default: return hipErrorInvalidValue;
    }
}
