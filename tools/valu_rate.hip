// Microbenchmark: issue rate of a few VALU instructions the LTS kernel leans on (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(int iters, double* out, double seed) {
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x * 1e-3;
    double w = seed * 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) v[i] = __builtin_fma(v[i], 1.0000001, w);
            else if (OP == 1) v[i] = fmin(v[i], v[(i + 1) & 15] + 0.0);      // min (plus an add to keep it honest)
            else if (OP == 2) v[i] = fmax(fmin(v[i], w), v[i] * 0.5);       // min + max + mul
            else if (OP == 3) { float a = (float)v[i]; a = fminf(a, (float)w); v[i] = a; }
            else if (OP == 4) v[i] = v[i] < w ? v[i] + 1.0 : v[i];          // cmp + cndmask + add
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, double* d) {
    const int iters = 4000, blocks = 256 * 4;          // 4 waves per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, 10, d, 1.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, d, 1.5);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x iters x 16 statements
    printf("%-28s %.3f ms -> %.2f ns per statement per SIMD\n", name, ms, ms * 1e6 / (4.0 * iters * 16));
}
int main() {
    double* d; (void)hipMalloc(&d, 4096 * 256 * sizeof(double));
    run<0>("fma f64", d);
    run<1>("min f64 (+add)", d);
    run<2>("min+max+mul f64", d);
    run<3>("cvt+min f32+cvt", d);
    run<4>("cmp+cndmask+add f64", d);
    return 0;
}
