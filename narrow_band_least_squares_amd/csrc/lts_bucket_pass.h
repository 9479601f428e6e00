// Device-side building blocks of the large-array FAST-LTS kernel's pair loops (solve_bucket.inc; also compiled on
// their own by tools/bk_pass_rate.hip, which times one pass in isolation).
#pragma once
#include <hip/hip_runtime.h>
#include "lts_bucket.h"

__device__ __forceinline__ unsigned int bk_key_hw(double r) {
    return (unsigned int)__double2hiint(r) & 0x7fffffffu;
}
__device__ __forceinline__ unsigned long long bk_key(double r) {
    return (unsigned long long)__double_as_longlong(r) & 0x7fffffffffffffffull;
}

// A block of 8 pairs' worth of the tables every lane needs: c0, c1 through the scalar cache (the compiler keeps
// them in SGPRs: uniform addresses of read-only global memory), y by LDS broadcast reads.  The k loops below fetch
// block n+1 before they work on block n, so that no pass waits for a load it has just issued (a pass is a chain of
// ~10 vector instructions per pair; with the loads inside the chain a pair cost ~130 cycles instead of ~40).
// The tables are padded by one block (plan: d_xs, d_xc; LDS: whatever follows y), so the fetch runs one block ahead
// without a bounds test.
struct BkBlk { double c[16]; double y[8]; };
__device__ __forceinline__ void bk_load(BkBlk& b, const double* __restrict__ xs, const double* y, int k0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) b.c[j] = xs[2 * k0 + j];
#pragma unroll
    for (int j = 0; j < 8; ++j) b.y[j] = y[k0 + j];
}
// Two blocks per trip, ping-pong (no register copies).  Per block: fetch the NEXT block first, compute the
// current block's bins (vector arithmetic only), then wait for everything outstanding and only THEN queue the
// block's eight histogram updates.  Scalar loads return out of order, so the one wait the hardware offers while one
// is in flight is lgkmcnt(0); placed here it only waits for operations that were queued a whole block of arithmetic
// earlier (the fetch, and the previous block's updates), and the updates it is followed by run beside the next
// block's arithmetic.  (With the updates in front of the wait every wave drained its own LDS atomics before it went
// on: with eight waves per CU the LDS and the vector unit took turns instead of overlapping — 240 cycles per pair.)
#define BK_LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f)
#define BK_BINS8(blk_, BINEXPR)                                                                \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                         \
        const double c0 = (blk_).c[2 * j_], c1 = (blk_).c[2 * j_ + 1], yk = (blk_).y[j_];      \
        const double r = (yk - c0 * z0) - c1 * z1;                                             \
        const unsigned int hw = bk_key_hw(r);                                                  \
        ad_[j_] = col + (BINEXPR) * RS;                                                       \
    }
#define BK_ADDS8()                                                                             \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_)                                           \
        __hip_atomic_fetch_add(ad_[j_], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#define BK_HIST_PASS(P_, xs_, y_, BINEXPR)                                                     \
    do {                                                                                       \
        const int nb_ = (P_) >> 3;                                                             \
        BkBlk A_, B_;                                                                          \
        unsigned int* ad_[8];                                                                  \
        bk_load(A_, xs_, y_, 0);                                                               \
        BK_LGKM0();                                                                            \
        int bi_ = 0;                                                                           \
        for (; bi_ + 1 < nb_; bi_ += 2) {                                                      \
            bk_load(B_, xs_, y_, (bi_ + 1) * 8);                                               \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_BINS8(A_, BINEXPR)                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_LGKM0();                                                                        \
            BK_ADDS8()                                                                         \
            bk_load(A_, xs_, y_, (bi_ + 2) * 8);                                               \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_BINS8(B_, BINEXPR)                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_LGKM0();                                                                        \
            BK_ADDS8()                                                                         \
        }                                                                                      \
        if (bi_ < nb_) { BK_BINS8(A_, BINEXPR) BK_ADDS8() }                                    \
        for (int k = nb_ * 8; k < (P_); ++k) {                                                 \
            const double r = ((y_)[k] - (xs_)[2 * k] * z0) - (xs_)[2 * k + 1] * z1;            \
            const unsigned int hw = bk_key_hw(r);                                              \
            __hip_atomic_fetch_add(col + (BINEXPR) * RS, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        }                                                                                      \
    } while (0)

// A plain pass over the pairs with the same fetch-ahead structure (the gather pass).
#define BK_BLOCK8(blk_, kbase_, BODY)                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                         \
        const double c0 = (blk_).c[2 * j_], c1 = (blk_).c[2 * j_ + 1], yk = (blk_).y[j_];      \
        BODY                                                                                   \
    }
#define BK_FOR_PAIRS8(P_, xs_, y_, BODY)                                                       \
    do {                                                                                       \
        const int nb_ = (P_) >> 3;                                                             \
        BkBlk A_, B_;                                                                          \
        bk_load(A_, xs_, y_, 0);                                                               \
        BK_LGKM0();                                                                            \
        int bi_ = 0;                                                                           \
        for (; bi_ + 1 < nb_; bi_ += 2) {                                                      \
            bk_load(B_, xs_, y_, (bi_ + 1) * 8);                                               \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_BLOCK8(A_, bi_ * 8, BODY)                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_LGKM0();                                                                        \
            bk_load(A_, xs_, y_, (bi_ + 2) * 8);                                               \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_BLOCK8(B_, (bi_ + 1) * 8, BODY)                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_LGKM0();                                                                        \
        }                                                                                      \
        if (bi_ < nb_) { BK_BLOCK8(A_, bi_ * 8, BODY) }                                        \
        for (int k = nb_ * 8; k < (P_); ++k) {                                                 \
            const double c0 = (xs_)[2 * k], c1 = (xs_)[2 * k + 1], yk = (y_)[k];               \
            BODY                                                                               \
        }                                                                                      \
    } while (0)

// The same for the sums pass: blocks of 4 pairs with c0, c1, c0*c1 in SGPRs and y from LDS; the other terms of the
// normal equations (c0^2, c1^2, c0 y, c1 y) are one vector multiply each — tables of them would not leave room for
// two blocks in the 100 SGPRs, or cost the LDS that a third workgroup per CU needs at 496 pairs.
struct BkBlkS { double c[8]; double cc[4]; double y[4]; };
__device__ __forceinline__ void bk_load_s(BkBlkS& b, const double* __restrict__ xs, const double* __restrict__ xc, const double* y, int k0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) b.c[j] = xs[2 * k0 + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) { b.cc[j] = xc[k0 + j]; b.y[j] = y[k0 + j]; }
}
#define BK_BLOCK4(blk_, kbase_, BODY)                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                         \
        const int k = (kbase_) + j_;                                                           \
        const double c0 = (blk_).c[2 * j_], c1 = (blk_).c[2 * j_ + 1], c01 = (blk_).cc[j_];    \
        const double yk = (blk_).y[j_];                                                        \
        BODY                                                                                   \
    }
#define BK_FOR_PAIRS_SUMS(P_, xs_, xc_, L_, BODY)                                              \
    do {                                                                                       \
        const int nb_ = (P_) >> 2;                                                             \
        BkBlkS A_, B_;                                                                         \
        bk_load_s(A_, xs_, xc_, (L_).y, 0);                                                    \
        BK_LGKM0();                                                                            \
        int bi_ = 0;                                                                           \
        for (; bi_ + 1 < nb_; bi_ += 2) {                                                      \
            bk_load_s(B_, xs_, xc_, (L_).y, (bi_ + 1) * 4);                                    \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_BLOCK4(A_, bi_ * 4, BODY)                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_LGKM0();                                                                        \
            bk_load_s(A_, xs_, xc_, (L_).y, (bi_ + 2) * 4);                                    \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_BLOCK4(B_, (bi_ + 1) * 4, BODY)                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            BK_LGKM0();                                                                        \
        }                                                                                      \
        if (bi_ < nb_) { BK_BLOCK4(A_, bi_ * 4, BODY) }                                        \
        for (int k = nb_ * 4; k < (P_); ++k) {                                                 \
            const double c0 = (xs_)[2 * k], c1 = (xs_)[2 * k + 1], c01 = (xc_)[k];             \
            const double yk = (L_).y[k];                                                       \
            BODY                                                                               \
        }                                                                                      \
    } while (0)

