// Microbenchmark (developer tool): issue cycles per wave of the vector instructions the LTS kernels are made of —
// which of them run at the full rate (4 cycles per wave64) on gfx950, and which do not.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(OP64)                                                                                          \
    asm volatile(OP64(0) OP64(1) OP64(2) OP64(3) OP64(4) OP64(5) OP64(6) OP64(7)                              \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c))

#define OP_MIN_F64(i) "v_min_f64 %" #i ", %" #i ", %8\n"
#define OP_MAX_F64(i) "v_max_f64 %" #i ", %" #i ", %8\n"
#define OP_ADD_F64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define OP_MUL_F64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define OP_FMA_F64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define OP_CMP_F64(i) "v_cmp_lt_f64 vcc, %" #i ", %8\n"
#define OP_CNDMASK2(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, double* sink, double bb, double cc, int reps) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = bb * (threadIdx.x + i);
    unsigned int u[8];
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 7 + i;
    const double b = bb, c = cc;
    const unsigned int ub = (unsigned int)bb + 5u, ub2 = (unsigned int)cc + 9u;
    const unsigned long long msk = 0x5555555555555555ull * (unsigned long long)(1 + (int)cc);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        if (MODE == 0) CHAIN8(OP_MIN_F64);
        if (MODE == 1) CHAIN8(OP_MAX_F64);
        if (MODE == 2) CHAIN8(OP_ADD_F64);
        if (MODE == 3) CHAIN8(OP_MUL_F64);
        if (MODE == 4) CHAIN8(OP_FMA_F64);
        if (MODE == 5) CHAIN8(OP_CMP_F64);
        if (MODE == 6) {
            asm volatile("v_min_u32 %0, %0, %8\nv_min_u32 %1, %1, %8\nv_min_u32 %2, %2, %8\nv_min_u32 %3, %3, %8\n"
                         "v_max_u32 %4, %4, %8\nv_max_u32 %5, %5, %8\nv_max_u32 %6, %6, %8\nv_max_u32 %7, %7, %8\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub));
        }
        if (MODE == 7) {
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub) : "vcc");
        }
        if (MODE == 8) {
            asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\n"
                         "v_lshrrev_b32 %4, 1, %4\nv_and_b32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_or_b32 %7, %7, %8\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub));
        }
        if (MODE == 9) {       // the mask in an SGPR pair that is not vcc
            asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\nv_cndmask_b32_e64 %1, %1, %8, %9\nv_cndmask_b32_e64 %2, %2, %8, %9\nv_cndmask_b32_e64 %3, %3, %8, %9\n"
                         "v_cndmask_b32_e64 %4, %4, %8, %9\nv_cndmask_b32_e64 %5, %5, %8, %9\nv_cndmask_b32_e64 %6, %6, %8, %9\nv_cndmask_b32_e64 %7, %7, %8, %9\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub), "s"(msk));
        }
        if (MODE == 10) {      // compare + select pairs, as a compiled `c ? a : b` on 64-bit values
            asm volatile("v_cmp_lt_f64 vcc, %4, %8\nv_cndmask_b32 %0, %0, %9, vcc\nv_cmp_lt_f64 vcc, %5, %8\nv_cndmask_b32 %1, %1, %9, vcc\n"
                         "v_cmp_lt_f64 vcc, %6, %8\nv_cndmask_b32 %2, %2, %9, vcc\nv_cmp_lt_f64 vcc, %7, %8\nv_cndmask_b32 %3, %3, %9, vcc\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b), "v"(ub) : "vcc");
        }
        if (MODE == 12) {      // one compare, seven selects on its vcc
            asm volatile("v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub) : "vcc");
        }
        if (MODE == 13) {      // a compiled 64-bit select: one compare, two selects
            asm volatile("v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cmp_lt_u32 vcc, %3, %8\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cmp_lt_u32 vcc, %6, %8\nv_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub) : "vcc");
        }
        if (MODE == 14) {      // vcc written by the scalar unit once per group
            asm volatile("s_mov_b64 vcc, %9\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub), "s"(msk) : "vcc");
        }
        if (MODE == 15) {      // compare, unrelated instruction, select
            asm volatile("v_cmp_lt_u32 vcc, %0, %8\nv_add_u32 %1, %1, %8\nv_cndmask_b32 %2, %2, %8, vcc\nv_add_u32 %3, %3, %8\n"
                         "v_cmp_lt_u32 vcc, %4, %8\nv_add_u32 %5, %5, %8\nv_cndmask_b32 %6, %6, %8, vcc\nv_add_u32 %7, %7, %8\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub) : "vcc");
        }
        if (MODE == 16) {      // selects on an old vcc, every other instruction
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_add_u32 %1, %1, %8\nv_cndmask_b32 %2, %2, %8, vcc\nv_add_u32 %3, %3, %8\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\nv_add_u32 %5, %5, %8\nv_cndmask_b32 %6, %6, %8, vcc\nv_add_u32 %7, %7, %8\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub) : "vcc");
        }
        if (MODE == 17) {      // the e64 encoding with vcc named as the mask
            asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\nv_cndmask_b32_e64 %1, %1, %8, vcc\nv_cndmask_b32_e64 %2, %2, %8, vcc\nv_cndmask_b32_e64 %3, %3, %8, vcc\n"
                         "v_cndmask_b32_e64 %4, %4, %8, vcc\nv_cndmask_b32_e64 %5, %5, %8, vcc\nv_cndmask_b32_e64 %6, %6, %8, vcc\nv_cndmask_b32_e64 %7, %7, %8, vcc\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub) : "vcc");
        }
        if (MODE == 18) {      // v_addc / v_subb chains also read vcc
            asm volatile("v_addc_co_u32 %0, vcc, %0, %8, vcc\nv_addc_co_u32 %1, vcc, %1, %8, vcc\nv_addc_co_u32 %2, vcc, %2, %8, vcc\nv_addc_co_u32 %3, vcc, %3, %8, vcc\n"
                         "v_addc_co_u32 %4, vcc, %4, %8, vcc\nv_addc_co_u32 %5, vcc, %5, %8, vcc\nv_addc_co_u32 %6, vcc, %6, %8, vcc\nv_addc_co_u32 %7, vcc, %7, %8, vcc\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub) : "vcc");
        }
        if (MODE == 19) {      // integer -> double conversions (the 0/1 weights of the LTS sums)
            asm volatile("v_cvt_f64_u32 %0, %8\nv_cvt_f64_u32 %1, %9\nv_cvt_f64_u32 %2, %10\nv_cvt_f64_u32 %3, %11\n"
                         "v_cvt_f64_u32 %4, %12\nv_cvt_f64_u32 %5, %13\nv_cvt_f64_u32 %6, %14\nv_cvt_f64_u32 %7, %15\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]));
        }
        if (MODE == 20) {      // v_rcp_f64 / v_sqrt_f64 (the quarter-rate unit) for comparison
            asm volatile("v_rcp_f64 %0, %0\nv_rcp_f64 %1, %1\nv_rcp_f64 %2, %2\nv_rcp_f64 %3, %3\n"
                         "v_rcp_f64 %4, %4\nv_rcp_f64 %5, %5\nv_rcp_f64 %6, %6\nv_rcp_f64 %7, %7\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        }
        if (MODE == 21) {      // v_bfe_u32 + v_cvt: the pair fit_sums issues per table entry
            asm volatile("v_bfe_u32 %0, %4, 3, 1\nv_bfe_u32 %1, %4, 4, 1\nv_bfe_u32 %2, %4, 5, 1\nv_bfe_u32 %3, %4, 6, 1\n"
                         "v_lshlrev_b32 %0, 20, %0\nv_lshlrev_b32 %1, 20, %1\nv_lshlrev_b32 %2, 20, %2\nv_lshlrev_b32 %3, 20, %3\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]) : "v"(ub));
        }
        if (MODE == 11) {      // v_cndmask with distinct destination registers (no chain through the destination)
            asm volatile("v_cndmask_b32 %0, %8, %9, vcc\nv_cndmask_b32 %1, %8, %9, vcc\nv_cndmask_b32 %2, %8, %9, vcc\nv_cndmask_b32 %3, %8, %9, vcc\n"
                         "v_cndmask_b32 %4, %8, %9, vcc\nv_cndmask_b32 %5, %8, %9, vcc\nv_cndmask_b32 %6, %8, %9, vcc\nv_cndmask_b32 %7, %8, %9, vcc\n"
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ub), "v"(ub2) : "vcc");
        }
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (double)u[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned long long* dout, double* sink) {
    const int reps = 1000, blocks = 512;
    for (int waves : {2, 4, 8}) {           // 256 CUs x 2 workgroups: one / two / four waves per SIMD
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(waves * 64), 0, 0, dout, sink, 1.25, 0.5, 10);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(waves * 64), 0, 0, dout, sink, 1.25, 0.5, reps * 20);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)reps * 20 * 64.0 * (waves / 2);        // instructions one SIMD issued
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(waves * 64), 0, 0, dout, sink, 1.25, 0.5, reps);
        (void)hipDeviceSynchronize();
        static unsigned long long h[512 * 8];
        (void)hipMemcpy(h, dout, sizeof(unsigned long long) * blocks * waves, hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < blocks * waves; ++i) mean += (double)h[i];
        mean /= blocks * waves;
        // s_memtime counts at a fixed 100 MHz; report the ratio to the 32-bit integer line instead of assuming a clock
        printf("%-34s %d waves per SIMD: %.3f counter ticks per instruction per wave | %.3f ns per instruction and SIMD (wall clock)\n", name, waves / 2, mean / (reps * 64.0), ms * 1e6 / per_simd);
    }
}

int main() {
    double* sink;
    unsigned long long* dout;
    (void)hipMalloc(&sink, 512 * 512 * sizeof(double));
    (void)hipMalloc(&dout, 512 * 8 * sizeof(unsigned long long));
    run<8>("32-bit integer mix", dout, sink);
    run<6>("v_min_u32 / v_max_u32", dout, sink);
    run<7>("v_cndmask_b32", dout, sink);
    run<0>("v_min_f64", dout, sink);
    run<1>("v_max_f64", dout, sink);
    run<2>("v_add_f64", dout, sink);
    run<3>("v_mul_f64", dout, sink);
    run<4>("v_fma_f64", dout, sink);
    run<5>("v_cmp_lt_f64", dout, sink);
    run<9>("v_cndmask_b32, mask in s[n:n+1]", dout, sink);
    run<11>("v_cndmask_b32, fresh destinations", dout, sink);
    run<12>("v_cmp + 7 v_cndmask on its vcc", dout, sink);
    run<13>("v_cmp + 2 v_cndmask (64-bit select)", dout, sink);
    run<14>("s_mov vcc + 7 v_cndmask", dout, sink);
    run<15>("v_cmp, v_add, v_cndmask, v_add", dout, sink);
    run<16>("v_cndmask (old vcc), v_add alternating", dout, sink);
    run<17>("v_cndmask_b32_e64 with vcc", dout, sink);
    run<18>("v_addc_co_u32 chain through vcc", dout, sink);
    run<19>("v_cvt_f64_u32", dout, sink);
    run<20>("v_rcp_f64", dout, sink);
    run<21>("v_bfe_u32 / v_lshlrev_b32", dout, sink);
    run<10>("v_cmp_lt_f64 + v_cndmask_b32 (8 instructions per group)", dout, sink);
    return 0;
}
