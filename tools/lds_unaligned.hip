// developer micro-benchmark: does ds_read_b64 accept addresses that are not 8-byte aligned on gfx950, does it return
// the right bytes, and at what rate?   hipcc --offload-arch=gfx950 -O3 tools/lds_unaligned.hip -o tools/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_check(int shift, unsigned long long* out) {
    __shared__ unsigned char buf[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)buf + 16 * threadIdx.x + shift;
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x] = v;
}

template <int STRIDE>
__global__ void k_rate(int shift, int iters, unsigned long long* out) {
    __shared__ unsigned char buf[65536];
    for (int i = threadIdx.x; i < 65536; i += blockDim.x) buf[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)buf + STRIDE * (threadIdx.x & 63) + shift;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned long long v0, v1, v2, v3;
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:1024\n\tds_read_b64 %2, %4 offset:2048\n\tds_read_b64 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr) : "memory");
        acc += v0 ^ v1 ^ v2 ^ v3;
    }
    if (acc == 0x1234) out[0] = acc;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8192);
    std::vector<unsigned long long> h(64);
    for (int shift = 0; shift < 8; ++shift) {
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, shift, d);
        if (hipDeviceSynchronize() != hipSuccess) { printf("shift %d: launch failed\n", shift); return 1; }
        hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int b = 0; b < 8; ++b) {
                const unsigned char want = (unsigned char)((16 * l + shift + b) * 7 + 3);
                if ((unsigned char)(h[l] >> (8 * b)) != want) ++bad;
            }
        printf("shift %d: %s (%d wrong bytes)\n", shift, bad ? "WRONG" : "ok", bad);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shift : {0, 4, 1, 3}) {
        const int iters = 20000;
        hipLaunchKernelGGL(k_rate<8>, dim3(256 * 4), dim3(256), 0, 0, shift, 10, d);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<8>, dim3(256 * 4), dim3(256), 0, 0, shift, iters, d);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 256.0 * 4 * 256 * iters * 4 * 8;
        printf("stride 8 B, shift %d: %.1f TB/s aggregate\n", shift, bytes / ms / 1e9);
    }
    return 0;
}
