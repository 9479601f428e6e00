// Microbenchmark (developer tool): CU-wide rate of LDS atomics against plain LDS stores, as a function of the waves
// that issue them — the histogram of the large-array FAST-LTS kernel (solve_bucket.inc) is one LDS update per
// (start, pair, pass).     hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_rate.hip -o tools/lds_atomic_rate
//   OP 0  ds_add_u32 (no return), conflict-free column layout [bin][32 words], random bins
//   OP 1  ds_write_b32, same addresses
//   OP 2  ds_add_u32, every lane its own word, fixed address (no address arithmetic at all)
//   OP 3  ds_add_rtn_u32 (returned value consumed)
//   OP 4  ds_add_u64
//   OP 5  no LDS operation (the address arithmetic alone)
//   OP 6  ds_add_u32, the same (random) row for every lane of the wave
//   OP 7  ds_add_u32, random rows, row stride 33 words
//   OP 8  ds_add_u32, random rows, 64-word rows, every lane its own column (u32 per lane)
//   OP 9  ds_add_u32, random rows, 16-word rows (four u8 counters per word: lanes l, l+16, l+32, l+48)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(512) void k(int iters, unsigned int* out, unsigned int seed) {
    extern __shared__ unsigned int sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned int* hist = sm + wv * (65 * 32 * 2);
    for (int i = lane; i < 65 * 32 * 2; i += 64) hist[i] = 0;
    __syncthreads();
    unsigned int x = seed + threadIdx.x * 2654435761u;
    unsigned int acc = 0;
    const unsigned int inc = lane < 32 ? 1u : 0x10000u;
    unsigned int* col = hist + (lane & 31);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u;
            const unsigned int b = (x >> 26);                  // 0..63
            if (OP == 0) __hip_atomic_fetch_add(col + b * 32, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 1) col[b * 32] = x;
            else if (OP == 2) __hip_atomic_fetch_add(hist + lane, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 3) acc += __hip_atomic_fetch_add(col + b * 32, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 4) __hip_atomic_fetch_add((unsigned long long*)(hist + 2 * (b * 32 + (lane & 31))), (unsigned long long)inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 6) __hip_atomic_fetch_add(col + (unsigned int)__builtin_amdgcn_readfirstlane((int)b) * 32, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 7) __hip_atomic_fetch_add(col + b * 33, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 8) __hip_atomic_fetch_add(hist + b * 64 + lane, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 9) __hip_atomic_fetch_add(hist + b * 16 + (lane & 15), 1u << (8 * (lane >> 4)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else acc += b;
        }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + hist[lane] + x;
}

template <int OP>
void run(const char* name, unsigned int* d, int waves_per_cu) {
    const int iters = 2000, blocks = 256;                                 // one workgroup per CU
    const int threads = waves_per_cu * 64;
    const size_t shm = (size_t)waves_per_cu * 65 * 32 * 2 * sizeof(unsigned int);
    (void)hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), shm, 0, 10, d, 7u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), shm, 0, iters, d, 7u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops_per_cu = (double)waves_per_cu * iters * 8;
    printf("%-34s waves/CU %d: %.3f ms -> %.1f ns per wave-op per CU (%.1f cycles at 2.1 GHz)\n", name, waves_per_cu, ms,
           ms * 1e6 / ops_per_cu, ms * 1e6 / ops_per_cu * 2.1);
}

int main() {
    unsigned int* d;
    (void)hipMalloc(&d, 256 * 512 * sizeof(unsigned int));
    for (int w : {4, 8}) {
        run<0>("ds_add_u32 column layout", d, w);
        run<1>("ds_write_b32 column layout", d, w);
        run<2>("ds_add_u32 fixed own word", d, w);
        run<3>("ds_add_rtn_u32 column layout", d, w);
        run<4>("ds_add_u64 column layout", d, w);
        run<5>("no LDS op", d, w);
        run<6>("ds_add_u32 same row all lanes", d, w);
        run<7>("ds_add_u32 row stride 33", d, w);
        run<8>("ds_add_u32 64-word rows", d, w);
        run<9>("ds_add_u32 16-word rows (u8)", d, w);
    }
    return 0;
}
