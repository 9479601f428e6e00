// Microbenchmark (r04): the K loop of the screening kernel in isolation, two tile shapes, same LDS footprint and occupancy
// as the kernel (512 threads, ~78 KB of LDS: two workgroups per CU, four waves per SIMD).
//   A  the shipped loop: four two-block tiles of v_mfma_i32_16x16x64_i8, 12 products per 64-sample K step,
//      6 KB of LDS reads per wave and step (NBLS_SCREEN_KLOOP_ASM, hand-scheduled)
//   B  32x32x32 tiles: two tiles that walk the A stream at one tile per K step (tile 1 at step n = tile 0 at step n + 1),
//      6 products per 32-sample K step, 4 KB of LDS reads per wave and step — the SAME multiply-adds per step as A
//      (compiler-scheduled; 64 accumulator registers)
// Question: does a third less LDS traffic per multiply-add buy matrix-pipe throughput?  ("what would move the
// screening kernel next".)  Prints issued int8 TOPS (2 ops per multiply-add) per variant.
//   hipcc -O3 --offload-arch=gfx950 -I narrow_band_least_squares_amd/csrc tools/kloop_rate.hip -o tools/kloop_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "screen_kloop.inc"
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int LDS_BYTES = 78 * 1024;
constexpr int CSA = 1440;           // copy stride of the kernel at W = 1200

__global__ __launch_bounds__(512, 4) void loop_a(int reps, int nst, int* out) {
    extern __shared__ unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 4; i += 512) ((int*)lds)[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    typedef const __attribute__((address_space(3))) unsigned char* lds_cp;
    const int g = lane >> 4;
    const unsigned char* Ah = lds + 28 * 1024 + (size_t)((wv >> 2) * 2) * 8 * CSA;
    const unsigned char* pAh = Ah + (size_t)(lane & 7) * CSA + 16 * g + (lane & 8);
    const unsigned char* pBh = lds + (size_t)((lane & 15) % 7) * 2 * 1792 + 16 * g;
    int acc = 0;
    for (int r = 0; r < reps; ++r) {
        v4i h0, m0, h1, m1, h2, m2, h3, m3;
        unsigned int va_h = (unsigned int)(uintptr_t)(lds_cp)pAh + 128 * (r & 3), vb_h = (unsigned int)(uintptr_t)(lds_cp)pBh;
        asm volatile("" : "+v"(va_h), "+v"(vb_h));
        unsigned int va_l = va_h + 8 * CSA, vb_l = vb_h + 1792;
        int kcnt;
        NBLS_SCREEN_KLOOP_ASM(h0, m0, h1, m1, h2, m2, h3, m3, va_h, va_l, vb_h, vb_l, nst, kcnt);
        acc += h0[0] + m0[1] + h1[2] + m1[3] + h2[0] + m2[1] + h3[2] + m3[3];
    }
    out[blockIdx.x * 512 + tid] = acc;
}

__device__ inline v4i ldf(const unsigned char* p) {
    typedef const volatile v2i __attribute__((address_space(3))) * lds_v2i;
    const v2i lo = *(lds_v2i)p;
    const v2i hi = *(lds_v2i)(p + 8);
    return (v4i){lo[0], lo[1], hi[0], hi[1]};
}

__global__ __launch_bounds__(512, 4) void loop_b(int reps, int nst32, int* out) {
    extern __shared__ unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 4; i += 512) ((int*)lds)[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    const int g = lane >> 5;                     // 32x32x32: lane = row + 32 * k-group, 16 bytes of K per lane
    const int r_ = lane & 31;
    const unsigned char* Ah = lds + 28 * 1024 + (size_t)((wv >> 2) * 2) * 8 * CSA;
    const unsigned char* pAh = Ah + (size_t)(r_ & 7) * CSA + 16 * g + (r_ & 24);
    const unsigned char* pAl = pAh + 8 * CSA;
    const unsigned char* pBh = lds + (size_t)((lane & 31) % 7) * 2 * 1792 + 16 * g;
    const unsigned char* pBl = pBh + 1792;
    int accs = 0;
    for (int r = 0; r < reps; ++r) {
        v16i H0 = {0}, M0 = {0}, H1 = {0}, M1 = {0};
        const unsigned char* qa_h = pAh + 128 * (r & 3);
        const unsigned char* qa_l = pAl + 128 * (r & 3);
        v4i f0h = ldf(qa_h), f0l = ldf(qa_l);
        for (int n = 0; n < nst32; ++n) {
            const v4i f1h = ldf(qa_h + 32 * (n + 1)), f1l = ldf(qa_l + 32 * (n + 1));
            const v4i bh = *(const v4i*)(pBh + 32 * n), bl = *(const v4i*)(pBl + 32 * n);
            H0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0h, bh, H0, 0, 0, 0);
            M0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0h, bl, M0, 0, 0, 0);
            M0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0l, bh, M0, 0, 0, 0);
            H1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f1h, bh, H1, 0, 0, 0);
            M1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f1h, bl, M1, 0, 0, 0);
            M1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f1l, bh, M1, 0, 0, 0);
            f0h = f1h; f0l = f1l;
        }
        accs += H0[0] + M0[1] + H1[2] + M1[3] + H0[15] + M1[15];
    }
    out[blockIdx.x * 512 + tid] = accs;
}

int main() {
    int* d;
    hipMalloc(&d, 4096 * 512 * sizeof(int));
    hipFuncSetAttribute((const void*)loop_a, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)loop_b, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int blocks = 512 * 4, reps = 400, nst = 19;          // 19 K steps of 64 samples = a lag group at lag 0 of a 1200-sample window
    for (int which = 0; which < 2; ++which) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(loop_a, dim3(blocks), dim3(512), LDS_BYTES, 0, reps, nst, d);
            else hipLaunchKernelGGL(loop_b, dim3(blocks), dim3(512), LDS_BYTES, 0, reps, 2 * nst, d);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waves = (double)blocks * 8;
        const double macs = waves * reps * (which == 0 ? nst * 12.0 * 16 * 16 * 64 : 2.0 * nst * 6.0 * 32 * 32 * 32);
        printf("%s: %.3f ms, %.0f issued int8 TOPS = %.3f of 5000 (two workgroups per CU, %d K steps per group)\n",
               which == 0 ? "A 16x16x64 x 4 two-block tiles, hand-scheduled, 6 KB LDS per step" : "B 32x32x32 x 2 tiles sharing the A stream, C++, 4 KB LDS per step",
               ms, 2.0 * macs / (ms * 1e-3) / 1e12, 2.0 * macs / (ms * 1e-3) / 1e12 / 5000.0, which == 0 ? nst : 2 * nst);
    }
    return 0;
}
