// Microbenchmark: issue rate of v_mfma_i32_16x16x64_i8 (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(int iters, int* out, int seed) {
    v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, 5, 7, 9};
    v4i c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = (v4i){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    int* d; hipMalloc(&d, 4096 * 256 * sizeof(int));
    // waves per SIMD: 1, 2, 4 (256-thread blocks = one wave per SIMD each), then 4 via bigger blocks
    for (int cfg = 0; cfg < 5; ++cfg) {
        const int wpb = cfg < 3 ? 256 : (cfg == 3 ? 512 : 1024);
        const int iters = 20000, blocks = cfg < 3 ? 256 * (1 << cfg) : 256 * (1024 / wpb);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(wpb), 0, 0, 100, d, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(wpb), 0, 0, iters, d, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waves = (double)blocks * wpb / 64, mfma = waves * iters * 16.0;
        printf("block %d threads, %d blocks: %.3f ms, %.2f ns per MFMA per wave-slot, %.1f TOPS (dense int8)\n", wpb, blocks, ms,
               ms * 1e6 / (iters * 16.0) , mfma * 16 * 16 * 64 * 2 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
