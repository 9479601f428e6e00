// MdCCM + slowness solve (ordinary least squares or FAST-LTS with reweighting) for every
// (band, window) unit of the plan.
//
// Replaces, inside lts_array.ltsva (narrow_band_least_squares.py:91,183; source absent,
// algorithm per SURVEY.md §3.3 / §8 rows a8-a12):
//   mdccm   = nanmedian_k(cmax_k)
//   OLS     : z = pinv(X) tau; sigma_tau = sqrt(tau.(tau - X z)/(P-2))
//   LTS     : y = tau / (1.4826 med|tau|), X = xij / xij_mad;
//             every elemental start -> exact fit -> <= csteps C-steps (h smallest |r|,
//             refit, stop when the objective repeats) -> ncand best distinct -> refine to
//             convergence -> best -> de-standardise -> raw scale -> weights -> WLS refit
//             -> reweighted scale -> final weights -> z, sigma_tau
//   vel = 1/||z||, baz = (atan2(z0, z1) deg - 360) mod 360
//
// One FAST-LTS start per lane; the 2x2 normal equations are solved per lane (no MFMA: 2x2
// Gramians); the per-window winner is found by a rank count in LDS.
//
// Arithmetic-order contract with oracle/nbls_oracle.py (this file is built with
// -ffp-contract=off): normal equations accumulated over k = 0..P-1 ascending, Cramer's
// rule, r_k = (y_k - x_k0 z0) - x_k1 z1, h-subset by stable rank of |r_k|, objective summed
// over the subset in ascending k.  With identical lags the LTS decisions are then identical.
#include "nbls_internal.h"
#include "lts_sortnet.inc"
#include "wave_ops.h"
#include "lts_bucket.h"
#include "lts_bucket_pass.h"
#include <utility>
#include <cstdlib>

namespace {

struct SArgs {
    const int32_t* lag;
    const double* cmax;
    int npairs;
    int vector_len;
    const int32_t* unit_off;
    const int32_t* win_off;   // [B] first window of each band's range (window sharding)
    const int32_t* unit_band;
    const int32_t* unit_win;  // [U] window index of every unit
    double fs;
    const double* xij;     // [P][2]
    const double* xpinv;   // [2][P]
    double* vel;
    double* baz;
    double* mdccm;
    double* sig;
    double* z;
    uint8_t* wts;
    // LTS
    const double* xs;      // [P][2]
    const int32_t* starts; // [S][4]
    int nstarts;
    int h;
    int csteps, csteps2, ncand;
    double xmad0, xmad1;
    double raw_factor;
    const double* rew;     // [P+1]
    double quantile;
    double zero_scale;
    int use_absr;
    int u0;                // first unit of this launch (unit batches of the pipelined path)
    int nsample;           // large-array kernel: selections per start centred by the sample histogram (option lts_sample_its)
    unsigned long long* stamps;   // developer (NBLS_LTS_STAMPS=1): 8 s_memtime stamps per wave, else NULL
    int stamp_waves;
    int stamp_mode;        // developer: 2 = the large-array kernel records every wave's busy cycles instead of thread 0's accounts
};

__device__ inline double dnan() { return __builtin_nan(""); }

__device__ inline void vel_baz(double z0, double z1, double* vel, double* baz) {
    *vel = 1.0 / sqrt(z0 * z0 + z1 * z1);
    const double x = atan2(z0, z1) * 180.0 / 3.141592653589793 - 360.0;
    double m = fmod(x, 360.0);
    if (m != 0.0) { if (m < 0.0) m += 360.0; } else m = 0.0;
    *baz = m;
}

// nanmedian of v[0..P) read from global memory by ONE thread (rank counting, no storage).
__device__ double nanmedian_serial(const double* v, int P) {
    int m = 0;
    for (int k = 0; k < P; ++k) m += (v[k] == v[k]);
    if (m == 0) return dnan();
    const int lo = (m - 1) / 2, hi = m / 2;
    double vlo = 0.0, vhi = 0.0;
    for (int k = 0; k < P; ++k) {
        const double vk = v[k];
        if (!(vk == vk)) continue;
        int rank = 0;
        for (int j = 0; j < P; ++j) {
            const double vj = v[j];
            rank += (vj < vk) || (vj == vk && j < k);
        }
        if (rank == lo) vlo = vk;
        if (rank == hi) vhi = vk;
    }
    return lo == hi ? vlo : (vlo + vhi) * 0.5;
}

// nanmedian of a lane's P values kept in LDS as column `lane` of col[k * 64 + lane] (rank counting; the serial form
// above reads global memory P^2 times per lane — 50-80 us per launch whatever the unit count, the largest stage of a
// small OLS call).
__device__ double nanmedian_lds(const double* col, int P) {
    int m = 0;
    for (int k = 0; k < P; ++k) { const double vk = col[k * 64]; m += (vk == vk); }
    if (m == 0) return dnan();
    const int lo = (m - 1) / 2, hi = m / 2;
    double vlo = 0.0, vhi = 0.0;
    for (int k = 0; k < P; ++k) {
        const double vk = col[k * 64];
        if (!(vk == vk)) continue;
        int rank = 0;
        for (int j = 0; j < P; ++j) {
            const double vj = col[j * 64];
            rank += (vj < vk) || (vj == vk && j < k);
        }
        if (rank == lo) vlo = vk;
        if (rank == hi) vhi = vk;
    }
    return lo == hi ? vlo : (vlo + vhi) * 0.5;
}

// ------------------------------------------------------------------------------------
// OLS: one lane per unit (64-lane workgroups; MdCCM through LDS when the pairs fit: smem_pairs = P, else 0).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void solve_ols_kernel(SArgs a, int nunits, int smem_pairs) {
    extern __shared__ double ols_col[];                   // [P][64]
    const int ul = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = ul < nunits;
    const int u = a.u0 + (live ? ul : 0);
    const int band = a.unit_band[u];
    const int w = u - a.unit_off[band] + a.win_off[band];   // window index inside the band (global)
    const int P = a.npairs;
    const int64_t o = (int64_t)band * a.vector_len + w;
    if (smem_pairs)
        for (int k = 0; k < P; ++k) ols_col[k * 64 + threadIdx.x] = a.cmax[o * P + k];
    if (!live) return;
    const int32_t* lag = a.lag + o * P;
    double z0 = 0.0, z1 = 0.0;
    for (int k = 0; k < P; ++k) {
        const double t = (double)lag[k] / a.fs;
        z0 = z0 + a.xpinv[k] * t;
        z1 = z1 + a.xpinv[P + k] * t;
    }
    double acc = 0.0;
    for (int k = 0; k < P; ++k) {
        const double t = (double)lag[k] / a.fs;
        const double r = t - (a.xij[2 * k] * z0 + a.xij[2 * k + 1] * z1);
        acc = acc + t * r;
        a.wts[o * P + k] = 1;
    }
    double vel, baz;
    vel_baz(z0, z1, &vel, &baz);
    a.vel[o] = vel;
    a.baz[o] = baz;
    a.sig[o] = sqrt(acc / (double)(P - 2));
    a.z[2 * o] = z0;
    a.z[2 * o + 1] = z1;
    a.mdccm[o] = smem_pairs ? nanmedian_lds(ols_col + threadIdx.x, P) : nanmedian_serial(a.cmax + o * P, P);
}

// ------------------------------------------------------------------------------------
// FAST-LTS: one workgroup (256 lanes) per unit, one elemental start per lane per round.
// ------------------------------------------------------------------------------------
constexpr int LT = 256;

struct Sel {
    double T;    // h-th smallest |r|
    int m;       // number of |r| == T members to take (in index order)
    double obj;  // sum of r^2 over the subset, ascending k
    bool ok;
};

__device__ inline double resid(const double* y, const double* X0, const double* X1, int k,
                               double z0, double z1) {
    return (y[k] - X0[k] * z0) - X1[k] * z1;
}

template <bool ABSR>
__device__ Sel select_h(const double* y, const double* X0, const double* X1, int P, int h,
                        double z0, double z1, double* absr, int tid) {
    Sel s;
    s.T = dnan();
    s.m = 0;
    s.obj = dnan();
    s.ok = false;
    if (ABSR) {
        for (int k = 0; k < P; ++k) absr[k * LT + tid] = fabs(resid(y, X0, X1, k, z0, z1));
    }
    int cl = 0;
    for (int k = 0; k < P; ++k) {
        const double ak = ABSR ? absr[k * LT + tid] : fabs(resid(y, X0, X1, k, z0, z1));
        int less = 0, eq = 0;
        for (int j = 0; j < P; ++j) {
            const double aj = ABSR ? absr[j * LT + tid] : fabs(resid(y, X0, X1, j, z0, z1));
            less += (aj < ak);
            eq += (aj == ak);
        }
        if (less <= h - 1 && h - 1 < less + eq) {
            s.T = ak;
            cl = less;
            s.ok = true;
        }
    }
    if (!s.ok) return s;
    s.m = h - cl;
    double obj = 0.0;
    int cnt = 0;
    for (int k = 0; k < P; ++k) {
        const double r = resid(y, X0, X1, k, z0, z1);
        const double ak = fabs(r);
        bool in = ak < s.T;
        if (ak == s.T && cnt < s.m) { in = true; ++cnt; }
        if (in) obj = obj + r * r;
    }
    s.obj = obj;
    return s;
}

// LS fit on the subset defined by (zold, T, m): normal equations, ascending k, Cramer.
__device__ inline void fit_subset(const double* y, const double* X0, const double* X1, int P,
                                  double zo0, double zo1, double T, int m, double* z0, double* z1) {
    double sxx = 0.0, sxy = 0.0, syy = 0.0, bx = 0.0, by = 0.0;
    int cnt = 0;
    for (int k = 0; k < P; ++k) {
        const double ak = fabs(resid(y, X0, X1, k, zo0, zo1));
        bool in = ak < T;
        if (ak == T && cnt < m) { in = true; ++cnt; }
        if (in) {
            const double x0 = X0[k], x1 = X1[k], yk = y[k];
            sxx = sxx + x0 * x0;
            sxy = sxy + x0 * x1;
            syy = syy + x1 * x1;
            bx = bx + x0 * yk;
            by = by + x1 * yk;
        }
    }
    const double det = sxx * syy - sxy * sxy;
    *z0 = (bx * syy - by * sxy) / det;
    *z1 = (by * sxx - bx * sxy) / det;
}

// LS fit on an explicit 0/1 mask (uint8 in LDS).
__device__ inline void fit_mask(const double* y, const double* X0, const double* X1, int P,
                                const uint8_t* mask, double* z0, double* z1) {
    double sxx = 0.0, sxy = 0.0, syy = 0.0, bx = 0.0, by = 0.0;
    for (int k = 0; k < P; ++k) {
        if (mask[k]) {
            const double x0 = X0[k], x1 = X1[k], yk = y[k];
            sxx = sxx + x0 * x0;
            sxy = sxy + x0 * x1;
            syy = syy + x1 * x1;
            bx = bx + x0 * yk;
            by = by + x1 * yk;
        }
    }
    const double det = sxx * syy - sxy * sxy;
    *z0 = (bx * syy - by * sxy) / det;
    *z1 = (by * sxx - bx * sxy) / det;
}

// Block-cooperative: sorted (ascending, NaN skipped) copy of v[0..P) into out; returns count.
__device__ int block_sort_small(const double* v, int P, double* out, int tid, int* cnt_sh) {
    if (tid == 0) *cnt_sh = 0;
    __syncthreads();
    for (int k = tid; k < P; k += LT) {
        const double vk = v[k];
        if (vk == vk) {
            int rank = 0;
            for (int j = 0; j < P; ++j) {
                const double vj = v[j];
                rank += (vj < vk) || (vj == vk && j < k);
            }
            out[rank] = vk;
            atomicAdd(cnt_sh, 1);
        }
    }
    __syncthreads();
    return *cnt_sh;
}

template <bool ABSR>
__global__ __launch_bounds__(LT) void solve_lts_kernel(SArgs a, int nunits) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x;
    const int u = a.u0 + blockIdx.x;
    const int band = a.unit_band[u];
    const int w = u - a.unit_off[band] + a.win_off[band];   // window index inside the band (global)
    const int P = a.npairs;
    const int S = a.nstarts;
    const int h = a.h;
    const int64_t o = (int64_t)band * a.vector_len + w;

    // ---- LDS carve-up ----
    double* tauv = sm;            // [P]
    double* y = tauv + P;         // [P]
    double* X0 = y + P;           // [P] standardised
    double* X1 = X0 + P;
    double* x0 = X1 + P;          // [P] original km
    double* x1 = x0 + P;
    double* tmp = x1 + P;         // [P] scratch (|tau|, cmax, residuals)
    double* srt = tmp + P;        // [P] sorted scratch
    double* objS = srt + P;       // [S]
    double* z0S = objS + S;       // [S]
    double* z1S = z0S + S;        // [S]
    double* cres = z1S + S;       // [NBLS_MAX_CAND][3]
    int* ord = (int*)(cres + 3 * NBLS_MAX_CAND);   // [S]
    int* rnk = ord + S;           // [P]
    int* cand = rnk + P;          // [NBLS_MAX_CAND]
    int* misc = cand + NBLS_MAX_CAND;              // [4]
    uint8_t* wsh = (uint8_t*)(misc + 4);           // [P]
    // absr (optional) behind, 8-byte aligned
    size_t off_bytes = (size_t)((uint8_t*)(wsh + P) - (uint8_t*)sm);
    off_bytes = (off_bytes + 7) & ~(size_t)7;
    double* absr = ABSR ? (double*)((uint8_t*)sm + off_bytes) : nullptr;

    for (int k = tid; k < P; k += LT) {
        const double t = (double)a.lag[o * P + k] / a.fs;
        tauv[k] = t;
        tmp[k] = fabs(t);
        X0[k] = a.xs[2 * k];
        X1[k] = a.xs[2 * k + 1];
        x0[k] = a.xij[2 * k];
        x1[k] = a.xij[2 * k + 1];
        wsh[k] = 1;
    }
    __syncthreads();
    // tmad = 1.4826 * median |tau|
    int m = block_sort_small(tmp, P, srt, tid, misc);
    const double med = (m & 1) ? srt[(m - 1) / 2] : (srt[m / 2 - 1] + srt[m / 2]) * 0.5;
    const double tmad = 1.4826 * med;
    __syncthreads();
    // MdCCM
    for (int k = tid; k < P; k += LT) tmp[k] = a.cmax[o * P + k];
    __syncthreads();
    m = block_sort_small(tmp, P, srt, tid, misc);
    if (tid == 0) {
        a.mdccm[o] = m == 0 ? dnan()
                            : ((m & 1) ? srt[(m - 1) / 2] : (srt[m / 2 - 1] + srt[m / 2]) * 0.5);
    }
    __syncthreads();

    if (tmad == 0.0) {   // "data spike" [R]: not processed
        if (tid == 0) {
            a.vel[o] = dnan(); a.baz[o] = dnan(); a.sig[o] = dnan();
            a.z[2 * o] = dnan(); a.z[2 * o + 1] = dnan();
        }
        for (int k = tid; k < P; k += LT) a.wts[o * P + k] = 1;
        return;
    }
    for (int k = tid; k < P; k += LT) y[k] = tauv[k] / tmad;
    __syncthreads();

    // ---- elemental starts, one per lane ----
    for (int s = tid; s < ((S + LT - 1) / LT) * LT; s += LT) {
        double obj = dnan(), z0 = dnan(), z1 = dnan();
        if (s < S) {
            const int i0 = a.starts[4 * s], i1 = a.starts[4 * s + 1];
            const int i2 = a.starts[4 * s + 2], i3 = a.starts[4 * s + 3];
            double sxx = 0.0, sxy = 0.0, syy = 0.0, bx = 0.0, by = 0.0;
            for (int k = 0; k < P; ++k) {
                if (k == i0 || k == i1 || k == i2 || k == i3) {
                    const double c0 = X0[k], c1 = X1[k], yk = y[k];
                    sxx = sxx + c0 * c0;
                    sxy = sxy + c0 * c1;
                    syy = syy + c1 * c1;
                    bx = bx + c0 * yk;
                    by = by + c1 * yk;
                }
            }
            const double det = sxx * syy - sxy * sxy;
            z0 = (bx * syy - by * sxy) / det;
            z1 = (by * sxx - bx * sxy) / det;
            Sel sel = select_h<ABSR>(y, X0, X1, P, h, z0, z1, absr, tid);
            bool active = sel.ok;
            double prev = 0.0;
            obj = active ? __builtin_inf() : dnan();
            for (int kk = 0; kk < a.csteps; ++kk) {
                if (!active) break;
                double n0, n1;
                fit_subset(y, X0, X1, P, z0, z1, sel.T, sel.m, &n0, &n1);
                sel = select_h<ABSR>(y, X0, X1, P, h, n0, n1, absr, tid);
                z0 = n0;
                z1 = n1;
                if (!sel.ok) { obj = dnan(); active = false; break; }
                obj = sel.obj;
                if (kk >= 1 && obj == prev) break;
                prev = obj;
            }
        }
        if (s < S) { objS[s] = obj; z0S[s] = z0; z1S[s] = z1; }
    }
    __syncthreads();

    // ---- rank the starts by (objective, start index); NaN/inf last ----
    for (int s = tid; s < S; s += LT) {
        const double os = objS[s];
        const bool fs_ = (os == os) && os < __builtin_inf();
        int rank = 0;
        for (int t = 0; t < S; ++t) {
            const double ot = objS[t];
            const bool ft = (ot == ot) && ot < __builtin_inf();
            bool before;
            if (ft && fs_) before = (ot < os) || (ot == os && t < s);
            else if (ft && !fs_) before = true;
            else if (!ft && fs_) before = false;
            else before = t < s;
            rank += before;
        }
        ord[rank] = s;
    }
    __syncthreads();
    if (tid == 0) {
        int nc = 0;
        for (int r = 0; r < S && nc < a.ncand; ++r) {
            const int s = ord[r];
            const double os = objS[s];
            if (!((os == os) && os < __builtin_inf())) break;
            bool dup = false;
            for (int c = 0; c < nc; ++c) {
                const int cs = cand[c];
                if (objS[cs] == os && z0S[cs] == z0S[s] && z1S[cs] == z1S[s]) { dup = true; break; }
            }
            if (!dup) cand[nc++] = s;
        }
        misc[1] = nc;
    }
    __syncthreads();
    const int nc = misc[1];
    // ---- refine the candidates to convergence, one per lane ----
    if (tid < nc) {
        double z0 = z0S[cand[tid]], z1 = z1S[cand[tid]];
        Sel sel = select_h<ABSR>(y, X0, X1, P, h, z0, z1, absr, tid);
        double pobj = 0.0, cobj = __builtin_inf();
        if (!sel.ok) cobj = dnan();
        else {
            for (int kk = 0; kk < a.csteps2; ++kk) {
                double n0, n1;
                fit_subset(y, X0, X1, P, z0, z1, sel.T, sel.m, &n0, &n1);
                sel = select_h<ABSR>(y, X0, X1, P, h, n0, n1, absr, tid);
                z0 = n0;
                z1 = n1;
                if (!sel.ok) { cobj = dnan(); break; }
                cobj = sel.obj;
                if (kk >= 1 && cobj == pobj) break;
                pobj = cobj;
            }
        }
        cres[3 * tid] = cobj;
        cres[3 * tid + 1] = z0;
        cres[3 * tid + 2] = z1;
    }
    __syncthreads();
    // ---- best candidate -> de-standardise -> residuals in original units ----
    double zr0 = dnan(), zr1 = dnan();
    {
        double best = __builtin_inf();
        for (int c = 0; c < nc; ++c) {
            if (cres[3 * c] < best) { best = cres[3 * c]; zr0 = cres[3 * c + 1]; zr1 = cres[3 * c + 2]; }
        }
        zr0 = zr0 * tmad / a.xmad0;
        zr1 = zr1 * tmad / a.xmad1;
    }
    const bool finite_z = (zr0 - zr0 == 0.0) && (zr1 - zr1 == 0.0);
    if (!finite_z) {
        if (tid == 0) {
            a.vel[o] = dnan(); a.baz[o] = dnan(); a.sig[o] = dnan();
            a.z[2 * o] = dnan(); a.z[2 * o + 1] = dnan();
        }
        for (int k = tid; k < P; k += LT) a.wts[o * P + k] = 1;
        return;
    }
    // stable rank of |r_k| (parallel over k)
    for (int k = tid; k < P; k += LT) tmp[k] = fabs(resid(tauv, x0, x1, k, zr0, zr1));
    __syncthreads();
    for (int k = tid; k < P; k += LT) {
        const double ak = tmp[k];
        int rank = 0;
        for (int j = 0; j < P; ++j) {
            const double aj = tmp[j];
            rank += (aj < ak) || (aj == ak && j < k);
        }
        rnk[k] = rank;
    }
    __syncthreads();
    if (tid == 0) {
        double ssq = 0.0;
        for (int k = 0; k < P; ++k) {
            if (rnk[k] < h) {
                const double r = resid(tauv, x0, x1, k, zr0, zr1);
                ssq = ssq + r * r;
            }
        }
        const double s0 = sqrt(ssq / (double)h) * a.raw_factor;
        double zf0 = zr0, zf1 = zr1;
        if (fabs(s0) < a.zero_scale) {
            for (int k = 0; k < P; ++k) wsh[k] = fabs(resid(tauv, x0, x1, k, zr0, zr1)) < a.zero_scale;
        } else {
            int nw = 0;
            for (int k = 0; k < P; ++k) {
                const double r = resid(tauv, x0, x1, k, zr0, zr1);
                const uint8_t wk = fabs(r / s0) <= a.quantile;
                wsh[k] = wk;
                nw += wk;
            }
            fit_mask(tauv, x0, x1, P, wsh, &zf0, &zf1);
            double ssw = 0.0;
            for (int k = 0; k < P; ++k) {
                if (wsh[k]) {
                    const double r = resid(tauv, x0, x1, k, zf0, zf1);
                    ssw = ssw + r * r;
                }
            }
            const double scale = nw > 1 ? sqrt(ssw / (double)(nw - 1)) * a.rew[nw] : 0.0;
            if (scale > 0.0) {
                for (int k = 0; k < P; ++k) {
                    const double r = resid(tauv, x0, x1, k, zf0, zf1);
                    wsh[k] = fabs(r / scale) <= a.quantile;
                }
            }
        }
        int nw = 0;
        double acc = 0.0;
        for (int k = 0; k < P; ++k) {
            if (wsh[k]) {
                ++nw;
                acc = acc + tauv[k] * resid(tauv, x0, x1, k, zf0, zf1);
            }
        }
        double vel, baz;
        vel_baz(zf0, zf1, &vel, &baz);
        a.vel[o] = vel;
        a.baz[o] = baz;
        a.sig[o] = nw > 2 ? sqrt(acc / (double)(nw - 2)) : dnan();
        a.z[2 * o] = zf0;
        a.z[2 * o + 1] = zf1;
    }
    __syncthreads();
    for (int k = tid; k < P; k += LT) a.wts[o * P + k] = wsh[k];
}


// ------------------------------------------------------------------------------------
// FAST-LTS, register-resident form for P known at compile time (P <= 64).
// One workgroup per unit, ONE elemental start per lane (blockDim = starts rounded up to a
// wave multiple, <= 512; more starts -> several rounds).  Each lane keeps its P absolute
// residuals in registers; the stable rank of every |r_k| is counted with fully unrolled
// compares (j < k: a_j <= a_k, j > k: a_j < a_k), giving the h-subset as a bit mask.
// The 2x2 normal equations are accumulated from per-unit product tables in LDS (broadcast
// reads).  The ncand best DISTINCT starts are peeled off by repeated workgroup arg-min with
// duplicate knock-out, refined one per lane, and lane 0 finishes (de-standardise, raw scale,
// weights, WLS refit, reweighting).  Same arithmetic order as the generic kernel / the oracle.
// ------------------------------------------------------------------------------------
template <int PT>
struct RegSel {
    unsigned long long mask;
    double obj;
    bool ok;
};

// branch-free "v if bit else +0.0" (the compiler turns a ?: around an LDS load into a branch)
__device__ inline double keep_if(double v, unsigned int bit) {
    const unsigned long long m = 0ull - (unsigned long long)bit;
    return __longlong_as_double((long long)((unsigned long long)__double_as_longlong(v) & m));
}

// Stable ranks of a[0..PT) with ONE compare per unordered pair: for j < k let c = (a_j <= a_k)
// (ties: lower index first).  Then rank_k = #{j<k: c_jk} + #{m>k: !c_km} = U_k + (PT-1-k) - L_k with
// U_k = sum_{j<k} c_jk and L_k = sum_{m>k} c_km.  Compile-time indices only (index_sequence folds),
// so a[], U[], L[] stay in registers.
template <int PT, int K, int... J>
__device__ inline void rank_pairs_k(const double (&a)[PT], int (&U)[PT], int (&L)[PT],
                                    std::integer_sequence<int, J...>) {
    auto one = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < K) {
            const int c = (a[j] <= a[K]) ? 1 : 0;
            U[K] += c;
            L[j] += c;
        }
    };
    (one(std::integral_constant<int, J>{}), ...);
}

template <int PT, int... K>
__device__ inline void find_threshold(const double (&a)[PT], int h, double& T, int& kstar,
                                      std::integer_sequence<int, K...>) {
    int U[PT], L[PT];
    ((U[K] = 0, L[K] = 0), ...);
    (rank_pairs_k<PT, K>(a, U, L, std::make_integer_sequence<int, PT>{}), ...);
    auto pick = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const int r = U[k] + (PT - 1 - k) - L[k];
        const bool hit = (r == h - 1) & (a[k] == a[k]);
        T = hit ? a[k] : T;
        kstar = hit ? k : kstar;
    };
    (pick(std::integral_constant<int, K>{}), ...);
}

template <int PT, int... K>
__device__ inline void residuals_reg(double (&a)[PT], const double* y, const double* X0, const double* X1,
                                     double z0, double z1, std::integer_sequence<int, K...>) {
    ((a[K] = fabs((y[K] - X0[K] * z0) - X1[K] * z1)), ...);
}

template <int PT, int... K>
__device__ inline void subset_reg(const double (&a)[PT], double T, int kstar, unsigned long long& mask,
                                  double& obj, std::integer_sequence<int, K...>) {
    auto one = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const unsigned int in = ((a[k] < T) | ((a[k] == T) & (k <= kstar))) ? 1u : 0u;
        obj = obj + keep_if(a[k] * a[k], in);
        mask |= (unsigned long long)in << k;
    };
    (one(std::integral_constant<int, K>{}), ...);
}

// Sorting networks for the pair counts of 4..8 element arrays (tools/gen_sortnet.py).
#define NBLS_CE(I, J)                              \
    {                                              \
        const double lo_ = fmin(v[I], v[J]);       \
        const double hi_ = fmax(v[I], v[J]);       \
        v[I] = lo_;                                \
        v[J] = hi_;                                \
    }
template <int PT> struct SortNet;
template <> struct SortNet<6> { static __device__ __forceinline__ void run(double (&v)[6]) { NBLS_SORTNET_6(NBLS_CE) } };
template <> struct SortNet<10> { static __device__ __forceinline__ void run(double (&v)[10]) { NBLS_SORTNET_10(NBLS_CE) } };
template <> struct SortNet<15> { static __device__ __forceinline__ void run(double (&v)[15]) { NBLS_SORTNET_15(NBLS_CE) } };
template <> struct SortNet<21> { static __device__ __forceinline__ void run(double (&v)[21]) { NBLS_SORTNET_21(NBLS_CE) } };
template <> struct SortNet<28> { static __device__ __forceinline__ void run(double (&v)[28]) { NBLS_SORTNET_28(NBLS_CE) } };
#undef NBLS_CE

template <int PT, int... K>
__device__ inline void copy_reg(double (&v)[PT], const double (&a)[PT], std::integer_sequence<int, K...>) {
    ((v[K] = a[K]), ...);
}

template <int PT, int... K>
__device__ inline void pick_sorted(const double (&v)[PT], int h, double& T, std::integer_sequence<int, K...>) {
    // h-1 >= PT/2 (h > P/2 for every alpha); wave-uniform compares
    ((T = (K >= PT / 2 && K == h - 1) ? v[K] : T), ...);
}

// h-subset for a threshold T that no tie straddles: in_k = (a_k <= T).  cnt tells the caller whether
// that holds (cnt == h).  obj as fit_sums: obj + a^2*w in one fma (a^2 rounded first).
template <int PT, int... K>
__device__ inline void subset_le(const double (&a)[PT], double T, unsigned int& mask, int& cnt, double& obj,
                                 std::integer_sequence<int, K...>) {
    auto one = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const unsigned int in = (a[k] <= T) ? 1u : 0u;
        obj = __builtin_fma(a[k] * a[k], (double)in, obj);
        mask |= in << k;
    };
    (one(std::integral_constant<int, K>{}), ...);
    cnt += __popc(mask);
}

// Ties across position h (cnt > h for the h-th smallest value T): the stable-rank subset is every a_k < T plus the
// FIRST h - #{a_k < T} of the a_k == T in index order.  On real lags this is not rare — closure (lag_ik = lag_ij +
// lag_jk, x_ik = x_ij + x_jk) makes r_ik = r_jk wherever a start's own pair has r_ij = 0 — so it must not cost the
// O(P^2) rank count: one equality mask, a few bit operations, and the objective again over the final subset.
template <int PT, int... K>
__device__ inline void eq_mask(const double (&a)[PT], double T, unsigned int& meq, std::integer_sequence<int, K...>) {
    ((meq |= (a[K] == T ? 1u : 0u) << K), ...);
}
template <int PT, int... K>
__device__ inline void obj_of_mask(const double (&a)[PT], unsigned int mask, double& obj, std::integer_sequence<int, K...>) {
    ((obj = __builtin_fma(a[K] * a[K], (double)((mask >> K) & 1u), obj)), ...);
}

template <int PT, int H = 0>     // H: the plan's h when it is known at compile time (0 = run-time h)
__device__ __forceinline__ RegSel<PT> select_reg(const double* y, const double* X0, const double* X1, int h,
                                        double z0, double z1) {
    using Seq = std::make_integer_sequence<int, PT>;
    double a[PT];
    // compiler barrier: without it LICM hoists every loop-invariant LDS table read of the C-step
    // loop into registers (hundreds of VGPRs) and the kernel spills
    asm volatile("" ::: "memory");
    residuals_reg<PT>(a, y, X0, X1, z0, z1, Seq{});
    RegSel<PT> s;
    s.mask = 0ull;
    s.obj = dnan();
    s.ok = false;
    // Usual case: the h-th smallest |r| from a sorting network (min/max pairs, no rank counting), the
    // subset by one compare per pair.  Same subset and objective as the stable-rank definition unless
    // a tie straddles position h (then cnt != h) or the fit is not finite: those lanes take the
    // rank-counting path below.
    if (fabs(z0) < 1.0e100 && fabs(z1) < 1.0e100) {
        double v[PT];
        copy_reg<PT>(v, a, Seq{});
        SortNet<PT>::run(v);
        // with h fixed at compile time only the comparators that feed output h-1 survive (P = 28,
        // h = 15: 235 of the 324 min/max instructions) and there is no run-time pick
        double T = v[H > 0 ? H - 1 : PT - 1];
        if (H == 0) pick_sorted<PT>(v, h, T, Seq{});
        unsigned int m = 0u;
        int cnt = 0;
        double obj = 0.0;
        subset_le<PT>(a, T, m, cnt, obj, Seq{});
        if (cnt == h) {
            s.ok = true;
            s.mask = (unsigned long long)m;
            s.obj = obj;
            return s;
        }
        if (cnt > h) {
            unsigned int meq = 0u;
            eq_mask<PT>(a, T, meq, Seq{});
            const unsigned int mlt = m & ~meq;
            const int need = h - __popc(mlt);            // 1 <= need < popc(meq)
            unsigned int rest = meq;
            for (int i = 0; i < need; ++i) rest &= rest - 1u;     // drop the lowest `need` tied indices
            const unsigned int mfin = mlt | (meq & ~rest);
            double o2 = 0.0;
            obj_of_mask<PT>(a, mfin, o2, Seq{});
            s.ok = true;
            s.mask = (unsigned long long)mfin;
            s.obj = o2;
            return s;
        }
    }
    double T = 0.0;
    int kstar = -1;
    find_threshold<PT>(a, h, T, kstar, Seq{});
    if (kstar < 0) return s;
    s.ok = true;
    double obj = 0.0;
    unsigned long long m = 0ull;
    subset_reg<PT>(a, T, kstar, m, obj, Seq{});
    s.mask = m;
    s.obj = obj;
    return s;
}

// Normal equations over the masked pairs from the product tables (ascending k), Cramer.
// sum + t*w with w = 0.0 / 1.0 as ONE fma: t*w is exact, so the fma rounds exactly like "sum + t" (or
// leaves the sum unchanged) -- same bits as a conditional add, one instruction per table entry.
template <int PT, int... K>
__device__ inline void fit_sums(const double* txx, const double* txy, const double* tyy, const double* tbx,
                                const double* tby, unsigned long long mask, double& sxx, double& sxy,
                                double& syy, double& bx, double& by, std::integer_sequence<int, K...>) {
    const unsigned int mlo = (unsigned int)mask, mhi = (unsigned int)(mask >> 32);
    auto one = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const unsigned int in = k < 32 ? ((mlo >> (k < 32 ? k : 0)) & 1u) : ((mhi >> (k >= 32 ? k - 32 : 0)) & 1u);
        const double w = (double)in;
        sxx = __builtin_fma(txx[k], w, sxx);
        sxy = __builtin_fma(txy[k], w, sxy);
        syy = __builtin_fma(tyy[k], w, syy);
        bx = __builtin_fma(tbx[k], w, bx);
        by = __builtin_fma(tby[k], w, by);
    };
    (one(std::integral_constant<int, K>{}), ...);
}

template <int PT>
__device__ __forceinline__ void fit_reg(const double* txx, const double* txy, const double* tyy, const double* tbx,
                               const double* tby, unsigned long long mask, double* z0, double* z1) {
    double sxx = 0.0, sxy = 0.0, syy = 0.0, bx = 0.0, by = 0.0;
    asm volatile("" ::: "memory");
    fit_sums<PT>(txx, txy, tyy, tbx, tby, mask, sxx, sxy, syy, bx, by, std::make_integer_sequence<int, PT>{});
    const double det = sxx * syy - sxy * sxy;
    *z0 = (bx * syy - by * sxy) / det;
    *z1 = (by * sxx - bx * sxy) / det;
}

// wave-synchronous hand-off through LDS: one wave owns its LDS slab, LDS ops of a wave execute
// in order, so only the compiler has to be kept from moving accesses across this point.
#define WSYNC()                                               \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

// median of the non-NaN entries of v[0..P) (P <= 64), computed by one wave: lane k ranks v[k].
__device__ inline double wave_nanmedian(const double* v, int P, double* srt, int lane) {
    const double vk = lane < P ? v[lane] : dnan();
    const bool ok = vk == vk;
    const int m = __popcll(__ballot(ok));
    if (ok) {
        int rank = 0;
        for (int j = 0; j < P; ++j) {
            const double vj = v[j];
            rank += (vj < vk) || (vj == vk && j < lane);
        }
        srt[rank] = vk;
    }
    WSYNC();
    double med = dnan();
    if (m > 0) med = (m & 1) ? srt[(m - 1) / 2] : (srt[m / 2 - 1] + srt[m / 2]) * 0.5;
    WSYNC();
    return med;
}

// FAST-LTS with ONE WAVE PER UNIT: the starts are swept in rounds of 64 lanes, the candidate
// peel-off / refinement / finish stay inside the wave (shuffles + the wave's own LDS slab), so
// there is no workgroup barrier and no idle wave during the serial tail.
template <int PT, int H>
__global__ __launch_bounds__(256, 2) void solve_lts_wave_kernel(SArgs a, int nunits, int slab_doubles) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int u = a.u0 + blockIdx.x * 4 + wv;
    if (u >= a.u0 + nunits) return;
    const int band = a.unit_band[u];
    const int w = a.unit_win[u];                            // window index inside the band (global)
    constexpr int P = PT;
    const int S = a.nstarts;
    const int h = H > 0 ? H : a.h;
    const int64_t o = (int64_t)band * a.vector_len + w;

    const int wave_id = blockIdx.x * 4 + wv;
#ifdef NBLS_DEVELOPER      // phase stamps exist in the developer build only
    unsigned long long* stp = (a.stamps && lane == 0 && wave_id < a.stamp_waves) ? a.stamps + (size_t)wave_id * 8 : nullptr;
#else
    unsigned long long* const stp = nullptr;
    (void)wave_id;
#endif
    if (stp) stp[0] = __builtin_amdgcn_s_memtime();
    double* base = sm + (size_t)wv * slab_doubles;
    double* tauv = base;
    double* y = tauv + P;
    double* X0 = y + P;
    double* X1 = X0 + P;
    double* x0 = X1 + P;
    double* x1 = x0 + P;
    double* txx = x1 + P;
    double* txy = txx + P;
    double* tyy = txy + P;
    double* tbx = tyy + P;
    double* tby = tbx + P;
    double* tmp = tby + P;
    double* srt = tmp + P;
    double* objS = srt + P;       // [S]
    double* z0S = objS + S;
    double* z1S = z0S + S;
    double* cres = z1S + S;       // [NBLS_MAX_CAND][3]
    int* cand = (int*)(cres + 3 * NBLS_MAX_CAND);  // [NBLS_MAX_CAND]
    uint8_t* wsh = (uint8_t*)(cand + NBLS_MAX_CAND);     // [P]
    // entry state (objS/z0S double as h-subset mask / previous objective while an entry is live)
    uint8_t* stt = wsh + P;                        // [S] 0 dead, 1 live, 2 finished, 3 knocked out, 4 merged
    unsigned short* act = (unsigned short*)(stt + S + ((S + P) & 1));   // [S] compact list of live entries

    if (lane < P) {
        const int k = lane;
        const double t = (double)a.lag[o * P + k] / a.fs;
        tauv[k] = t;
        tmp[k] = fabs(t);
        X0[k] = a.xs[2 * k];
        X1[k] = a.xs[2 * k + 1];
        x0[k] = a.xij[2 * k];
        x1[k] = a.xij[2 * k + 1];
        wsh[k] = 1;
    }
    WSYNC();
    const double tmad = 1.4826 * wave_nanmedian(tmp, P, srt, lane);
    if (lane < P) tmp[lane] = a.cmax[o * P + lane];
    WSYNC();
    const double md = wave_nanmedian(tmp, P, srt, lane);
    if (lane == 0) a.mdccm[o] = md;
    if (tmad == 0.0) {     // "data spike" [R]: not processed
        if (lane == 0) {
            a.vel[o] = dnan(); a.baz[o] = dnan(); a.sig[o] = dnan();
            a.z[2 * o] = dnan(); a.z[2 * o + 1] = dnan();
        }
        if (lane < P) a.wts[o * P + lane] = 1;
        return;
    }
    if (lane < P) {
        const int k = lane;
        const double yk = tauv[k] / tmad;
        const double c0 = X0[k], c1 = X1[k];
        y[k] = yk;
        txx[k] = c0 * c0;
        txy[k] = c0 * c1;
        tyy[k] = c1 * c1;
        tbx[k] = c0 * yk;
        tby[k] = c1 * yk;
    }
    WSYNC();

    // ---- elemental starts: exact fit -> h-subset; starts that land on the SAME h-subset continue
    //      identically from here on (a C-step depends on the subset only), so only the first of each
    //      group is carried through the C-steps: typically 378 starts -> ~140 distinct subsets after the
    //      initial fit and a few dozen after the first C-step.  The merged starts would have finished
    //      with the same (objective, z) and be knocked out as duplicates by the candidate peel anyway.
    if (stp) stp[1] = __builtin_amdgcn_s_memtime();
    unsigned long long* maskS = (unsigned long long*)objS;      // live entry: current h-subset
    double* prevS = z0S;                                        // live entry: previous objective
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        if (s < S) {
            unsigned long long sm_ = 0ull;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = a.starts[4 * s + q];
                if (idx >= 0) sm_ |= 1ull << idx;
            }
            double z0, z1;
            const int i0 = a.starts[4 * s], i1 = a.starts[4 * s + 1];
            if (a.starts[4 * s + 2] < 0 && i0 >= 0 && i1 >= 0 && i0 != i1) {
                // two-point start: the sums have two terms (0 + t_i) + t_j, the same bits in either order
                const double sxx = txx[i0] + txx[i1], sxy = txy[i0] + txy[i1], syy = tyy[i0] + tyy[i1];
                const double bx = tbx[i0] + tbx[i1], by = tby[i0] + tby[i1];
                const double det = sxx * syy - sxy * sxy;
                z0 = (bx * syy - by * sxy) / det;
                z1 = (by * sxx - bx * sxy) / det;
            } else {
                fit_reg<PT>(txx, txy, tyy, tbx, tby, sm_, &z0, &z1);
            }
            const RegSel<PT> sel = select_reg<PT, H>(y, X0, X1, h, z0, z1);
            maskS[s] = sel.mask;
            prevS[s] = 0.0;
            stt[s] = sel.ok ? 1 : 0;
            act[s] = (unsigned short)s;
        }
    }
    WSYNC();
    if (stp) stp[2] = __builtin_amdgcn_s_memtime();
    int nact = S;
    for (int kk = 0; kk <= a.csteps; ++kk) {
        // -- rebuild the live list: drop dead/final entries and entries whose subset an earlier live
        //    entry already has (the list is ordered by start index, so the first of a group survives)
        if (kk <= 1) {
            // duplicates are found with a one-probe hash table of start indices in LDS (z1S is free until
            // the first entry finishes, which cannot happen before the second C-step): slot = min index
            // of the entries hashing there; an entry is dropped only if the slot's entry has the SAME
            // subset, so a hash collision merely leaves a mergeable entry in the list (harmless)
            unsigned int* tab = (unsigned int*)z1S;
            int hbits = 4;
            while ((2 << hbits) * 4 <= S * 8 && hbits < 10) ++hbits;      // table bytes <= z1S bytes
            const int HS = 1 << hbits;
            for (int q = lane; q < HS; q += 64) tab[q] = 0xffffffffu;
            WSYNC();
            for (int p0 = 0; p0 < nact; p0 += 64) {
                const int pos = p0 + lane;
                if (pos < nact) {
                    const int id = act[pos];
                    if (stt[id] == 1) {
                        const unsigned long long mk = maskS[id];
                        const unsigned int hs = ((unsigned int)(mk ^ (mk >> 29)) * 0x9E3779B1u) >> (32 - hbits);
                        atomicMin(&tab[hs], (unsigned int)id);
                    }
                }
            }
            WSYNC();
            for (int p0 = 0; p0 < nact; p0 += 64) {
                const int pos = p0 + lane;
                if (pos < nact) {
                    const int id = act[pos];
                    if (stt[id] == 1) {
                        const unsigned long long mk = maskS[id];
                        const unsigned int hs = ((unsigned int)(mk ^ (mk >> 29)) * 0x9E3779B1u) >> (32 - hbits);
                        const unsigned int w = tab[hs];
                        if (w != (unsigned int)id && maskS[w] == mk) stt[id] = 4;     // merged into entry w
                    }
                }
            }
            WSYNC();
        }
        int nn = 0;
        for (int p0 = 0; p0 < nact; p0 += 64) {
            const int pos = p0 + lane;
            const int id = pos < nact ? act[pos] : 0;
            const bool keep = pos < nact && stt[id] == 1;
            const unsigned long long bal = __ballot(keep);
            const int dst = nn + __popcll(bal & ((1ull << lane) - 1ull));
            WSYNC();
            if (keep) act[dst] = (unsigned short)id;
            nn += __popcll(bal);
            WSYNC();
        }
        nact = nn;
        if (kk == a.csteps || nact == 0) break;
        // -- one C-step for every live entry
        for (int p0 = 0; p0 < nact; p0 += 64) {
            const int pos = p0 + lane;
            if (pos < nact) {
                const int id = act[pos];
                const unsigned long long mk = maskS[id];
                const double prev = prevS[id];
                double n0, n1;
                fit_reg<PT>(txx, txy, tyy, tbx, tby, mk, &n0, &n1);
                const RegSel<PT> sel = select_reg<PT, H>(y, X0, X1, h, n0, n1);
                if (!sel.ok) {
                    stt[id] = 0;
                } else if ((kk >= 1 && sel.obj == prev) || kk == a.csteps - 1) {
                    objS[id] = sel.obj;          // finished: converged, or the last allowed C-step
                    z0S[id] = n0;
                    z1S[id] = n1;
                    stt[id] = 2;
                } else {
                    maskS[id] = sel.mask;
                    prevS[id] = sel.obj;
                }
            }
        }
        WSYNC();
    }
    WSYNC();

    // ---- peel off the ncand best distinct starts ----
    if (stp) stp[3] = __builtin_amdgcn_s_memtime();
    int nc = 0;
    // finished entries, compacted in start order (the live list is no longer needed)
    int nfin = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool keep = s < S && stt[s] == 2;
        const unsigned long long bal = __ballot(keep);
        if (keep) act[nfin + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)s;
        nfin += __popcll(bal);
    }
    WSYNC();
    if (stp) stp[7] = (unsigned long long)nfin;
    if (nfin <= 128) {
        // up to two finished entries per lane, the whole peel in registers: wave minimum of the
        // objective; the lowest list position holding it is the lowest start index (slot 0 of every lane
        // comes before slot 1); its duplicates (same objective and z) are dropped with it
        int id[2];
        double ov[2], w0[2], w1[2];
        bool alive[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const bool have = lane + 64 * r < nfin;
            id[r] = have ? act[lane + 64 * r] : 0;
            ov[r] = have ? objS[id[r]] : dnan();
            w0[r] = have ? z0S[id[r]] : 0.0;
            w1[r] = have ? z1S[id[r]] : 0.0;
            alive[r] = have;
        }
        for (int it = 0; it < a.ncand; ++it) {
            const double c0 = alive[0] ? ov[0] : __builtin_inf(), c1 = alive[1] ? ov[1] : __builtin_inf();
            const double m = nbls_wave::min_f64(fmin(c0, c1));
            if (!(m < __builtin_inf())) break;       // nothing alive with a finite objective
            const unsigned long long win0 = __ballot(alive[0] && ov[0] == m);
            const unsigned long long win1 = __ballot(alive[1] && ov[1] == m);
            double ww0, ww1;
            int wid;
            if (win0) {
                const int wl = __builtin_ctzll(win0);
                ww0 = nbls_wave::readlane_f64(w0[0], wl); ww1 = nbls_wave::readlane_f64(w1[0], wl);
                wid = __builtin_amdgcn_readlane(id[0], wl);
            } else {
                const int wl = __builtin_ctzll(win1);
                ww0 = nbls_wave::readlane_f64(w0[1], wl); ww1 = nbls_wave::readlane_f64(w1[1], wl);
                wid = __builtin_amdgcn_readlane(id[1], wl);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (alive[r] && ov[r] == m && w0[r] == ww0 && w1[r] == ww1) alive[r] = false;
            if (lane == 0) cand[nc] = wid;
            ++nc;
        }
    } else
    for (int it = 0; it < a.ncand; ++it) {
        double bv = __builtin_inf();
        int bs = 0x7fffffff;
        for (int s = lane; s < S; s += 64) {
            if (stt[s] == 2) {
                const double ov = objS[s];
                if (ov < bv || (ov == bv && s < bs)) { bv = ov; bs = s; }
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off, 64);
            const int os = __shfl_xor(bs, off, 64);
            if (ov < bv || (ov == bv && os < bs)) { bv = ov; bs = os; }
        }
        if (bs == 0x7fffffff) break;      // nothing alive (wave-uniform after the butterfly)
        const double w0 = z0S[bs], w1 = z1S[bs];
        for (int s = lane; s < S; s += 64)
            if (stt[s] == 2 && objS[s] == bv && z0S[s] == w0 && z1S[s] == w1) stt[s] = 3;
        if (lane == 0) cand[nc] = bs;
        ++nc;
        WSYNC();
    }
    WSYNC();
    // ---- refine, one candidate per lane ----
    if (stp) stp[4] = __builtin_amdgcn_s_memtime();
    if (lane < nc) {
        double z0 = z0S[cand[lane]], z1 = z1S[cand[lane]];
        RegSel<PT> sel = select_reg<PT, H>(y, X0, X1, h, z0, z1);
        double pobj = 0.0, cobj = __builtin_inf();
        if (!sel.ok) cobj = dnan();
        else {
            for (int kk = 0; kk < a.csteps2; ++kk) {
                double n0, n1;
                fit_reg<PT>(txx, txy, tyy, tbx, tby, sel.mask, &n0, &n1);
                sel = select_reg<PT, H>(y, X0, X1, h, n0, n1);
                z0 = n0;
                z1 = n1;
                if (!sel.ok) { cobj = dnan(); break; }
                cobj = sel.obj;
                if (kk >= 1 && cobj == pobj) break;
                pobj = cobj;
            }
        }
        cres[3 * lane] = cobj;
        cres[3 * lane + 1] = z0;
        cres[3 * lane + 2] = z1;
    }
    WSYNC();
    if (stp) stp[5] = __builtin_amdgcn_s_memtime();
    // ---- finish: best candidate -> de-standardise -> raw scale -> weights -> WLS refit -> reweighting.
    //      Uniform values are computed by every lane (same cost as one lane); lane k < P owns pair k for the
    //      residuals and the divisions of the weight tests; the order-sensitive sums stay sequential in k.
    {
        double zr0 = dnan(), zr1 = dnan();
        double best = __builtin_inf();
        for (int c = 0; c < nc; ++c)
            if (cres[3 * c] < best) { best = cres[3 * c]; zr0 = cres[3 * c + 1]; zr1 = cres[3 * c + 2]; }
        zr0 = zr0 * tmad / a.xmad0;
        zr1 = zr1 * tmad / a.xmad1;
        const bool finite_z = (zr0 - zr0 == 0.0) && (zr1 - zr1 == 0.0);
        if (!finite_z) {
            if (lane == 0) {
                a.vel[o] = dnan(); a.baz[o] = dnan(); a.sig[o] = dnan();
                a.z[2 * o] = dnan(); a.z[2 * o + 1] = dnan();
            }
            if (lane < P) a.wts[o * P + lane] = wsh[lane];
        } else {
            const int k = lane < P ? lane : 0;
            const double tk = tauv[k], c0 = x0[k], c1 = x1[k];
            const unsigned long long pmask = (P < 64) ? ((1ull << P) - 1ull) : ~0ull;
            RegSel<PT> sel = select_reg<PT, H>(tauv, x0, x1, h, zr0, zr1);
            const double s0 = sqrt(sel.obj / (double)h) * a.raw_factor;
            double zf0 = zr0, zf1 = zr1;
            const double rk0 = (tk - c0 * zr0) - c1 * zr1;
            unsigned long long wm;
            if (fabs(s0) < a.zero_scale) {
                wm = __ballot(fabs(rk0) < a.zero_scale) & pmask;
            } else {
                wm = __ballot(fabs(rk0 / s0) <= a.quantile) & pmask;
                int nw = __popcll(wm);
                // product tables in original units (the C-step tables are no longer needed)
                WSYNC();
                if (lane < P) {
                    txx[k] = c0 * c0;
                    txy[k] = c0 * c1;
                    tyy[k] = c1 * c1;
                    tbx[k] = c0 * tk;
                    tby[k] = c1 * tk;
                }
                WSYNC();
                fit_reg<PT>(txx, txy, tyy, tbx, tby, wm, &zf0, &zf1);
                const double rk = (tk - c0 * zf0) - c1 * zf1;
                if (lane < P) tmp[k] = rk * rk;
                WSYNC();
                double ssw = 0.0;
                for (int q = 0; q < P; ++q) ssw = __builtin_fma(tmp[q], (double)(unsigned int)((wm >> q) & 1ull), ssw);
                const double scale = nw > 1 ? sqrt(ssw / (double)(nw - 1)) * a.rew[nw] : 0.0;
                if (scale > 0.0) wm = __ballot(fabs(rk / scale) <= a.quantile) & pmask;
                WSYNC();
            }
            const double rkf = (tk - c0 * zf0) - c1 * zf1;
            if (lane < P) tmp[k] = tk * rkf;
            WSYNC();
            const int nw = __popcll(wm);
            double acc = 0.0;
            for (int q = 0; q < P; ++q) acc = __builtin_fma(tmp[q], (double)(unsigned int)((wm >> q) & 1ull), acc);
            if (lane == 0) {
                double vel, baz;
                vel_baz(zf0, zf1, &vel, &baz);
                a.vel[o] = vel;
                a.baz[o] = baz;
                a.sig[o] = nw > 2 ? sqrt(acc / (double)(nw - 2)) : dnan();
                a.z[2 * o] = zf0;
                a.z[2 * o + 1] = zf1;
            }
            if (lane < P) a.wts[o * P + lane] = (uint8_t)((wm >> lane) & 1ull);
        }
    }
    if (stp) stp[6] = __builtin_amdgcn_s_memtime();
}

int lts_wave_slab_doubles(int P, int S) {
    size_t b = (size_t)(13 * P + 3 * S + 3 * NBLS_MAX_CAND) * sizeof(double);
    b += (size_t)NBLS_MAX_CAND * sizeof(int);
    b += (size_t)P;                     // wsh
    b += (size_t)S + 2;                 // entry state (+ alignment of the u16 list)
    b += (size_t)S * sizeof(unsigned short);
    b = (b + 15) & ~(size_t)15;
    return (int)(b / sizeof(double));
}

template <int PT, int H>
hipError_t launch_fast_h(nbls_handle* h, const SArgs& a, int nunits, hipStream_t st) {
    int slab = lts_wave_slab_doubles(PT, a.nstarts);
    slab += h->opt.lts_pad_kb * 128;                                              // developer: occupancy experiment
    const size_t shm = (size_t)slab * 4 * sizeof(double);
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)solve_lts_wave_kernel<PT, H>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((solve_lts_wave_kernel<PT, H>), dim3((nunits + 3) / 4), dim3(256), shm, st, a, nunits, slab);
    return hipGetLastError();
}

// HALF = h of alpha = 0.5 for this pair count (the default LTS setting): its own instantiation with the
// selection network pruned to that output; every other h runs the generic one.
template <int PT, int HALF>
hipError_t launch_fast(nbls_handle* h, const SArgs& a, int nunits, hipStream_t st) {
    const bool generic_only = h->opt.lts_generic_h != 0;
    if (a.h == HALF && !generic_only) return launch_fast_h<PT, HALF>(h, a, nunits, st);
    return launch_fast_h<PT, 0>(h, a, nunits, st);
}

#include "solve_bucket.inc"

// ------------------------------------------------------------------------------------
// Confidence intervals of the slowness estimate — ltsva's 7th / 8th returns (vel_uncert, baz_uncert at
// narrow_band_least_squares.py:91, example.py:109; Szuberla & Olson 2004 as used by lts_array [R]): the 90 %
// confidence ellipse of the slowness vector, semi-axes sqrt(chi2_{0.90,2}) sigma_tau / sqrt(lambda_i) along the
// eigenvectors of X^T X, centred on z.  vel_uncert = half the spread of 1/|s| over the ellipse, baz_uncert = half the
// angle the ellipse subtends at the origin (NaN when the origin is inside).  One lane per unit, behind the unit's solve:
//   radial extrema   stationary points of |c + (a cos phi, b sin phi)|^2: the best of NPHI boundary samples (cos / sin
//                    table shared through LDS), then NEWT clipped Newton steps on f'(phi) = 0
//   subtended angle  the two tangents from the origin in closed form (unit circle after scaling by the semi-axes)
// Same operations in the same order as oracle/nbls_oracle.py: confidence_intervals_closed_form (un-fused arithmetic:
// this file is built with -ffp-contract=off).
// ------------------------------------------------------------------------------------
struct UArgs {
    const double* z;          // [B][VL][2]
    const double* sig;        // [B][VL]
    double* vunc;             // [B][VL]
    double* bunc;             // [B][VL]
    const int32_t* unit_band;
    const int32_t* unit_win;
    int vector_len, u0, nunits;
    double ev0, ev1;          // eigenvalues of X^T X (ascending, numpy.linalg.eigh)
    double r00, r01, r10, r11;// rotation into the eigen-frame
};
constexpr int UNC_NPHI = 720;
constexpr int UNC_NEWT = 8;
constexpr double UNC_CHI2 = 4.605170185988092;          // -2 ln(1 - 0.90) = chi2.ppf(0.90, 2)

__device__ inline double py_mod360(double x) {          // Python's x % 360.0
    double m = fmod(x, 360.0);
    if (m != 0.0) { if (m < 0.0) m += 360.0; } else m = 0.0;
    return m;
}

__global__ __launch_bounds__(64) void uncertainty_kernel(UArgs a) {
    __shared__ double cs[UNC_NPHI], sn[UNC_NPHI];
    const double step = 6.283185307179586 / (double)UNC_NPHI;
    for (int k = threadIdx.x; k < UNC_NPHI; k += 64) {
        double s_, c_;
        sincos((double)k * step, &s_, &c_);
        sn[k] = s_; cs[k] = c_;
    }
    __syncthreads();
    const int ul = blockIdx.x * 64 + threadIdx.x;
    if (ul >= a.nunits) return;
    const int u = a.u0 + ul;
    const int64_t o = (int64_t)a.unit_band[u] * a.vector_len + a.unit_win[u];
    const double z0 = a.z[2 * o], z1 = a.z[2 * o + 1], sg = a.sig[o];
    const double nan_ = dnan();
    const double q = sqrt(UNC_CHI2);
    const double sa = q * sg / sqrt(a.ev0), sb = q * sg / sqrt(a.ev1);
    const double x0 = z0 * a.r00 + z1 * a.r01, y0 = z0 * a.r10 + z1 * a.r11;
    const bool ok = isfinite(sa) && isfinite(sb) && isfinite(x0) && isfinite(y0);
    if (!ok) { a.vunc[o] = nan_; a.bunc[o] = nan_; return; }
    if (sa == 0.0 && sb == 0.0) { a.vunc[o] = 0.0; a.bunc[o] = 0.0; return; }     // exact fit: a point, no spread
    // ---- radial extrema ----
    int kmin = 0, kmax = 0;
    double fmin_ = 0.0, fmax_ = 0.0;
    for (int k = 0; k < UNC_NPHI; ++k) {
        const double cx = x0 + sa * cs[k], cy = y0 + sb * sn[k];
        double f = cx * cx + cy * cy;
        if (!isfinite(f)) f = 0.0;
        if (k == 0 || f < fmin_) { fmin_ = f; kmin = k; }
        if (k == 0 || f > fmax_) { fmax_ = f; kmax = k; }
    }
    double rext[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        double p = (double)(e == 0 ? kmin : kmax) * step;
        for (int it = 0; it < UNC_NEWT; ++it) {
            double s_, c_;
            sincos(p, &s_, &c_);
            const double g = -sa * s_ * (x0 + sa * c_) + sb * c_ * (y0 + sb * s_);                               // f'/2
            const double c2 = c_ * c_ - s_ * s_;
            const double hh = -sa * c_ * x0 - sa * sa * c2 - sb * s_ * y0 + sb * sb * c2;                        // f''/2
            double st = fabs(hh) > 0.0 ? g / hh : 0.0;
            st = st < -0.05 ? -0.05 : (st > 0.05 ? 0.05 : st);
            p = p - st;
        }
        double s_, c_;
        sincos(p, &s_, &c_);
        rext[e] = hypot(x0 + sa * c_, y0 + sb * s_);
    }
    a.vunc[o] = 0.5 * fabs(1.0 / rext[0] - 1.0 / rext[1]);
    // ---- subtended angle ----
    const double pz = sa > 0.0 ? x0 / sa : __builtin_inf(), qz = sb > 0.0 ? y0 / sb : __builtin_inf();
    const double d2 = pz * pz + qz * qz;
    if (!(d2 > 1.0)) { a.bunc[o] = nan_; return; }                                  // origin inside: direction undetermined
    const double root = sqrt(d2 - 1.0) / d2, kk = 1.0 - 1.0 / d2;
    const double t1x = sa * (pz * kk - root * qz), t1y = sb * (qz * kk + root * pz);
    const double t2x = sa * (pz * kk + root * qz), t2y = sb * (qz * kk - root * pz);
    // back to the east / north frame: t @ R
    const double e1x = t1x * a.r00 + t1y * a.r10, e1y = t1x * a.r01 + t1y * a.r11;
    const double e2x = t2x * a.r00 + t2y * a.r10, e2y = t2x * a.r01 + t2y * a.r11;
    const double th1 = py_mod360(atan2(e1y, e1x) * (180.0 / 3.141592653589793) - 360.0);
    const double th2 = py_mod360(atan2(e2y, e2x) * (180.0 / 3.141592653589793) - 360.0);
    double dth = fabs(th1 - th2);
    if (dth > 180.0) dth = fabs(dth - 360.0);
    a.bunc[o] = 0.5 * dth;
}

size_t lts_lds_bytes(int P, int S, bool absr) {
    size_t b = (size_t)(8 * P + 3 * S + 3 * NBLS_MAX_CAND) * sizeof(double);
    b += (size_t)(S + P + NBLS_MAX_CAND + 4) * sizeof(int);
    b += (size_t)P;
    b = (b + 7) & ~(size_t)7;
    if (absr) b += (size_t)P * LT * sizeof(double);
    return b;
}

}  // namespace

hipError_t nbls_launch_solve(nbls_handle* h) { return nbls_launch_solve_range(h, 0, h->nunits, h->stream); }

// LTS weights u8[B][VL][P] -> bit mask u8[B][VL][MB] in the result block (bit k & 7 of byte k >> 3 = pair k).
// One thread per (unit, mask byte).  Rows of windows that were not computed stay zero (memset by execute).
__global__ __launch_bounds__(256) void pack_weights_kernel(const uint8_t* wts, uint8_t* mask, const int32_t* unit_band,
                                                           const int32_t* unit_win, int vector_len, int P, int MB, int u0,
                                                           int nunits) {
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= nunits * MB) return;
    const int ul = item / MB, mb = item - ul * MB;
    const int u = u0 + ul;
    const int64_t o = (int64_t)unit_band[u] * vector_len + unit_win[u];
    const uint8_t* w = wts + o * P + 8 * mb;
    const int n = P - 8 * mb < 8 ? P - 8 * mb : 8;
    unsigned int m = 0;
    for (int k = 0; k < n; ++k) m |= (unsigned int)(w[k] != 0) << k;
    mask[o * MB + mb] = (uint8_t)m;
}

static hipError_t solve_range_impl(nbls_handle* h, int64_t u0, int64_t nu, hipStream_t st);

hipError_t nbls_launch_pack_weights(nbls_handle* h, int64_t u0, int64_t nu, hipStream_t st) {
    if (nu <= 0) return hipSuccess;
    const int MB = h->mask_bytes;
    const int64_t items = nu * MB;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, h->d_wts, h->d_mask,
                       h->d_unit_band, h->d_unit_win, h->vector_len, h->npairs, MB, (int)u0, (int)nu);
    return hipGetLastError();
}

hipError_t nbls_launch_solve_range(nbls_handle* h, int64_t u0, int64_t nu, hipStream_t st) {
    hipError_t e = solve_range_impl(h, u0, nu, st);
    if (e != hipSuccess) return e;
    if (h->want_unc && h->d_unc && nu > 0) {
        UArgs a{};
        const size_t cells = (size_t)h->nbands * h->vector_len;
        a.z = h->d_z; a.sig = h->d_sig; a.vunc = h->d_unc; a.bunc = h->d_unc + cells;
        a.unit_band = h->d_unit_band; a.unit_win = h->d_unit_win;
        a.vector_len = h->vector_len; a.u0 = (int)u0; a.nunits = (int)nu;
        a.ev0 = h->unc_par[0]; a.ev1 = h->unc_par[1];
        a.r00 = h->unc_par[2]; a.r01 = h->unc_par[3]; a.r10 = h->unc_par[4]; a.r11 = h->unc_par[5];
        hipLaunchKernelGGL(uncertainty_kernel, dim3((unsigned)((nu + 63) / 64)), dim3(64), 0, st, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return nbls_launch_pack_weights(h, u0, nu, st);
}

static hipError_t solve_range_impl(nbls_handle* h, int64_t u0, int64_t nu, hipStream_t st) {
    if (nu <= 0) return hipSuccess;
    SArgs a{};
    a.u0 = (int)u0;
    a.lag = h->d_lag;
    a.cmax = h->d_cmax;
    a.npairs = h->npairs;
    a.vector_len = h->vector_len;
    a.unit_off = h->d_unit_off;
    a.win_off = h->d_win_off;
    a.unit_band = h->d_unit_band;
    a.unit_win = h->d_unit_win;
    a.fs = h->fs;
    a.xij = h->d_xij;
    a.xpinv = h->d_xpinv;
    a.vel = h->d_vel;
    a.baz = h->d_baz;
    a.mdccm = h->d_mdccm;
    a.sig = h->d_sig;
    a.z = h->d_z;
    a.wts = h->d_wts;
    const int nunits = (int)nu;
    if (!h->lts) {
        const int sp = h->npairs <= 64 ? h->npairs : 0;          // the lanes' MdCCM columns in LDS (<= 32 KB)
        hipLaunchKernelGGL(solve_ols_kernel, dim3((nunits + 63) / 64), dim3(64), (size_t)sp * 64 * sizeof(double), st, a, nunits, sp);
        return hipGetLastError();
    }
    a.xs = h->d_xs;
    a.starts = h->d_starts;
    a.nstarts = h->ltsp.nstarts;
    a.h = h->ltsp.h;
    a.csteps = h->ltsp.csteps;
    a.csteps2 = h->ltsp.csteps2;
    a.ncand = h->ltsp.ncand;
    a.xmad0 = h->ltsp.xij_mad[0];
    a.xmad1 = h->ltsp.xij_mad[1];
    a.raw_factor = h->ltsp.raw_factor;
    a.rew = h->d_rew;
    a.quantile = h->ltsp.quantile;
    a.zero_scale = h->ltsp.zero_scale;
    a.nsample = h->opt.lts_sample_its > 0 ? h->opt.lts_sample_its : (h->opt.lts_sample_its < 0 ? 0 : 1000);
    a.stamps = nullptr;
    a.stamp_waves = 0;
    {
        if (h->opt.lts_stamps && h->d_stamps) {
            a.stamps = h->d_stamps;
            const int64_t cap = (int64_t)(h->cap_stamps / (8 * sizeof(unsigned long long))) / 2;   // second half: the cooperative kernel's C-step breakdown
            a.stamp_waves = (int)(nunits < cap ? nunits : cap);
            h->lts_stamp_waves = a.stamp_waves;
            a.stamp_mode = h->opt.lts_stamps;
        }
    }
    if (h->opt.lts_impl != 1) {
        switch (h->npairs) {     // register-resident kernel for 4..8 elements (larger P spills registers)
            case 6: return launch_fast<6, 4>(h, a, nunits, st);
            case 10: return launch_fast<10, 6>(h, a, nunits, st);
            case 15: return launch_fast<15, 9>(h, a, nunits, st);
            case 21: return launch_fast<21, 12>(h, a, nunits, st);
            case 28: return launch_fast<28, 15>(h, a, nunits, st);
            default: break;
        }
    }
    if (h->opt.lts_impl == 0 && h->npairs <= NBLS_MAX_PAIRS && a.nstarts <= NBLS_MAX_STARTS && h->d_xc) {
        // one start per lane, bucket selection (solve_bucket.inc): every other pair count (9..32 elements).
        // Up to 255 pairs: u8 histogram counters and merging of identical subsets (three 4-wave workgroups per CU at
        // 120 pairs); beyond: u16 counters and no merging (it removes a tenth of the starts at 496 pairs and its masks
        // are what would keep a CU at ONE workgroup: two per CU let one unit's refinement run beside the next one's
        // C-steps).
        const bool small = h->npairs <= 255;
        int threads = 256;
        if (h->opt.lts_coop_threads >= 64 && h->opt.lts_coop_threads <= 512) threads = h->opt.lts_coop_threads & ~63;
        const size_t bshm = lts_bucket_lds_bytes(h->npairs, a.nstarts, threads / 64, small ? 4 : 2, small);
        if (bshm <= 160 * 1024) {
            const void* fn = small ? (const void*)solve_lts_bucket_kernel<4, true> : (const void*)solve_lts_bucket_kernel<2, false>;
            hipError_t be = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bshm);
            if (be != hipSuccess) return be;
            if (small)
                hipLaunchKernelGGL((solve_lts_bucket_kernel<4, true>), dim3(nunits), dim3(threads), bshm, st, a, (const double*)h->d_xs, (const double*)h->d_xc, (const double*)h->d_xss, nunits);
            else
                hipLaunchKernelGGL((solve_lts_bucket_kernel<2, false>), dim3(nunits), dim3(threads), bshm, st, a, (const double*)h->d_xs, (const double*)h->d_xc, (const double*)h->d_xss, nunits);
            return hipGetLastError();
        }
    }
    // cache |r_k| per lane in LDS when a workgroup's slab fits in half a CU's LDS
    const bool absr = lts_lds_bytes(h->npairs, a.nstarts, true) <= 80 * 1024;
    a.use_absr = absr;
    const size_t shm = lts_lds_bytes(h->npairs, a.nstarts, absr);
    if (shm > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e;
    if (absr) {
        e = hipFuncSetAttribute((const void*)solve_lts_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((solve_lts_kernel<true>), dim3(nunits), dim3(LT), shm, st, a, nunits);
    } else {
        e = hipFuncSetAttribute((const void*)solve_lts_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((solve_lts_kernel<false>), dim3(nunits), dim3(LT), shm, st, a, nunits);
    }
    return hipGetLastError();
}
