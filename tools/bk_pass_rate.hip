// Microbenchmark (developer tool): cycles per pair of ONE histogram pass / ONE sums pass of the large-array FAST-LTS
// kernel, in isolation, with parts taken out — which of {vector arithmetic, LDS atomics, LDS reads, scalar loads}
// a pass really waits for.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I narrow_band_least_squares_amd/csrc tools/bk_pass_rate.hip -o tools/bk_pass_rate
// MODE 0 full pass | 1 no histogram update | 2 y from a register instead of LDS | 3 c0, c1 from registers instead of
// the scalar cache | 4 arithmetic only | 5 the sums pass
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lts_bucket_pass.h"

struct FakeL { const double* y; };

template <int MODE>
__global__ __launch_bounds__(512) void k(const double* __restrict__ xs, const double* __restrict__ xc, const double* yg, int P, int reps,
                                         unsigned long long* out, double* sink) {
    extern __shared__ unsigned int sm[];
    constexpr int ROW = 16, RS = 16;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    double* y = (double*)sm;                       // [3 * (P + 16)]
    unsigned int* hist = (unsigned int*)(y + 3 * (P + 16)) + wv * 65 * RS;
    for (int i = threadIdx.x; i < 3 * (P + 16); i += blockDim.x) y[i] = yg[i % P];
    for (int i = lane; i < 65 * RS; i += 64) hist[i] = 0;
    __syncthreads();
    const double z0 = 0.3 + 1e-3 * threadIdx.x, z1 = -0.2 + 1e-3 * lane;
    const unsigned int lo_hw = 0x3ff00000u - (32u << 18);
    const int sh = 18;
    const unsigned int inc = 1u << (8 * (lane / ROW));
    unsigned int* col = hist + (lane & (ROW - 1));
    double acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 0) {
            BK_HIST_PASS(P, xs, y, nbls_bucket::bin_of_hw(lo_hw, sh, hw));
        } else if (MODE == 5) {
            FakeL L{y};
            double obj = 0, sxx = 0, sxy = 0, syy = 0, bx = 0, by = 0;
            unsigned int mw = 0;
            const unsigned long long T = 0x3ff0000000000000ull;
            BK_FOR_PAIRS_SUMS(P, xs, xc, L, {
                const double r = (yk - c0 * z0) - c1 * z1;
                const bool in = bk_key(r) < T;
                const double w = in ? 1.0 : 0.0;
                obj = __builtin_fma(r * r, w, obj);
                sxx = __builtin_fma(c0 * c0, w, sxx);
                sxy = __builtin_fma(c01, w, sxy);
                syy = __builtin_fma(c1 * c1, w, syy);
                bx = __builtin_fma(c0 * yk, w, bx);
                by = __builtin_fma(c1 * yk, w, by);
                mw = (mw << 1) | (unsigned int)in;
                if ((k & 31) == 31) { hist[lane & 15] = mw; mw = 0u; }
            });
            acc += obj + sxx + sxy + syy + bx + by;
        } else {
            // the same arithmetic, written plainly, with one ingredient replaced at a time
            double ya = y[lane & 7], ca = xs[lane & 7], cb = xs[8 + (lane & 7)];
            unsigned int asum = 0;
#pragma unroll 8
            for (int kk = 0; kk < P; ++kk) {
                const double yk = (MODE == 2 || MODE == 4) ? ya : y[kk];
                const double c0 = (MODE == 3 || MODE == 4) ? ca : xs[2 * kk];
                const double c1 = (MODE == 3 || MODE == 4) ? cb : xs[2 * kk + 1];
                const double r = (yk - c0 * z0) - c1 * z1;
                const int b = nbls_bucket::bin_of_hw(lo_hw, sh, bk_key_hw(r));
                if (MODE == 1 || MODE == 4) asum += (unsigned int)b;
                else __hip_atomic_fetch_add(col + b * RS, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ya += 1e-9; ca += 1e-9;
            }
            acc += (double)asum;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * nw + wv] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + (double)hist[lane];
}

template <int MODE>
void run(const char* name, const double* xs, const double* xc, const double* yg, int P, int waves, unsigned long long* dout, double* sink) {
    const int reps = 50, blocks = 256;
    const size_t shm = (size_t)3 * (P + 16) * 8 + (size_t)waves * 65 * 16 * 4;
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(waves * 64), shm, 0, xs, xc, yg, P, 2, dout, sink);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(waves * 64), shm, 0, xs, xc, yg, P, reps, dout, sink);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 8];
    (void)hipMemcpy(h, dout, sizeof(unsigned long long) * blocks * waves, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (int i = 0; i < blocks * waves; ++i) { mean += (double)h[i]; mx = mx > (double)h[i] ? mx : (double)h[i]; }
    mean /= blocks * waves;
    printf("%-44s P %d waves/CU %d: %.1f cycles (s_memtime) per pair per wave, slowest wave %.1f\n", name, P, waves,
           mean / (reps * (double)P), mx / (reps * (double)P));
}

int main() {
    const int P = 496;
    double h[3 * (512 + 16)];
    for (int i = 0; i < 3 * (512 + 16); ++i) h[i] = 0.37 * ((i * 2654435761u) % 1000) / 1000.0 - 0.2;
    double *xs, *xc, *yg, *sink;
    unsigned long long* dout;
    (void)hipMalloc(&xs, sizeof(h));
    (void)hipMalloc(&xc, sizeof(h));
    (void)hipMalloc(&yg, sizeof(h));
    (void)hipMalloc(&sink, 256 * 512 * sizeof(double));
    (void)hipMalloc(&dout, 256 * 8 * sizeof(unsigned long long));
    (void)hipMemcpy(xs, h, sizeof(h), hipMemcpyHostToDevice);
    (void)hipMemcpy(xc, h, sizeof(h), hipMemcpyHostToDevice);
    (void)hipMemcpy(yg, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w : {1, 4, 8}) {
        run<0>("histogram pass (kernel form)", xs, xc, yg, P, w, dout, sink);
        run<1>("plain loop, no histogram update", xs, xc, yg, P, w, dout, sink);
        run<2>("plain loop, y from a register", xs, xc, yg, P, w, dout, sink);
        run<3>("plain loop, c0 c1 from registers", xs, xc, yg, P, w, dout, sink);
        run<4>("plain loop, arithmetic only", xs, xc, yg, P, w, dout, sink);
        run<5>("sums pass (kernel form)", xs, xc, yg, P, w, dout, sink);
    }
    return 0;
}
