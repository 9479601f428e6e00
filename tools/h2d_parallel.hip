// Developer probe: 8 pageable rows of 6.9 MB host -> device, one stream against two and four streams driven by as many host
// threads (the runtime pins the pages of a pageable source per copy: do two copies in flight hide each other's pinning?).
//   hipcc -O2 --offload-arch=gfx950 -o tools/h2d_parallel tools/h2d_parallel.hip -pthread && ./tools/h2d_parallel
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int nrows = 8;
    const size_t npts = argc > 1 ? (size_t)atol(argv[1]) : 864000, bytes = npts * 8;
    std::vector<double*> rows(nrows);
    for (int r = 0; r < nrows; ++r) { rows[r] = (double*)malloc(bytes); for (size_t i = 0; i < npts; ++i) rows[r][i] = (double)(i + r); }
    double* d = nullptr;
    CK(hipMalloc((void**)&d, bytes * nrows));
    hipStream_t st[4];
    for (int i = 0; i < 4; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    for (int nthr : {1, 2, 4, 1, 2, 4}) {
        double best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            auto work = [&](int t) {
                (void)hipSetDevice(0);
                for (int r = t; r < nrows; r += nthr) (void)hipMemcpyAsync(d + (size_t)r * npts, rows[r], bytes, hipMemcpyHostToDevice, st[t]);
                (void)hipStreamSynchronize(st[t]);
            };
            for (int t = 1; t < nthr; ++t) th.emplace_back(work, t);
            work(0);
            for (auto& x : th) x.join();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best) best = ms;
        }
        printf("%d thread(s) / stream(s): %.3f ms for %.1f MB = %.1f GB/s\n", nthr, best, bytes * nrows / 1e6, bytes * nrows / best / 1e6);
    }
    return 0;
}
