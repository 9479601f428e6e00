// TEST INFRASTRUCTURE, not part of the product: a loopback stand-in for the ten RCCL entry points that
// csrc/comm.hip resolves with dlopen, for ONE process whose "ranks" are several library handles on the SAME GPU.
// It lets a one-GPU box run the whole multi-rank host and library path — per-rank launch threads, result blocks in
// HBM, nbls_comm_gather with its stand-in blocks and status words, the root's assembly — with device-to-device
// copies where RCCL would move the blocks over xGMI.  It says nothing about RCCL or xGMI themselves.
// A second form serves ranks that are separate PROCESSES on that one GPU (the launcher form: python -m
// torch.distributed.run --nproc-per-node 2 ...): ncclCommInitRank with a world > 1 maps a file under /dev/shm named
// after the unique id, and the all-gather goes device -> shared host memory -> device with arrive / depart counters.
// Selected with nbls_comm_set_library(<this .so>, 1) (dist.set_transport_library; bench.py --transport-lib).
//   hipcc -O2 -shared -fPIC tests/c_caller/loopback_rccl.cpp -o tests/c_caller/libloopback_rccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
// ranks in separate processes: the mapped file
constexpr int MAXR = 16;
constexpr size_t SLOT = (size_t)64 << 20;            // bytes per rank (sparse: only what is written gets pages)
struct Shm {
    std::atomic<int> arrive[MAXR], depart[MAXR];
    char pad[4096 - 2 * MAXR * sizeof(std::atomic<int>)];
    char data[1];                                    // MAXR slots of SLOT bytes follow
};
struct Group { int n; };
struct Comm { Group* g; int rank, n; Shm* shm = nullptr; int gen = 0; };
double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
bool wait_all(std::atomic<int>* a, int n, int gen) {
    const double t0 = now_s();
    for (int q = 0; q < n; ++q)
        while (a[q].load(std::memory_order_acquire) < gen) {
            if (now_s() - t0 > 120.0) return false;  // a peer died: report, do not hang
            usleep(50);
        }
    return true;
}
// all-gather across processes: device -> my slot, wait for everyone, all slots -> device
ncclResult_t shm_allgather(Comm* c, const void* sbuf, void* rbuf, size_t bytes, hipStream_t st) {
    if (bytes > SLOT) return ncclInvalidArgument;
    Shm* m = c->shm;
    const int gen = ++c->gen;
    if (!wait_all(m->depart, c->n, gen - 1)) return ncclSystemError;       // nobody still reads the previous round
    if (hipMemcpyAsync(m->data + (size_t)c->rank * SLOT, sbuf, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    m->arrive[c->rank].store(gen, std::memory_order_release);
    if (!wait_all(m->arrive, c->n, gen)) return ncclSystemError;
    for (int q = 0; q < c->n; ++q)
        if (hipMemcpyAsync((char*)rbuf + (size_t)q * bytes, m->data + (size_t)q * SLOT, bytes, hipMemcpyHostToDevice, st) != hipSuccess)
            return ncclUnhandledCudaError;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    m->depart[c->rank].store(gen, std::memory_order_release);
    return ncclSuccess;
}
struct Op { int kind; const void* sbuf; void* rbuf; size_t bytes; int peer; Comm* c; hipStream_t st; };   // 0 send, 1 recv, 2 all-gather
std::mutex mu;
std::vector<Op> ops;
int depth = 0;

size_t tsize(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

// dst (on stream rst) <- src (produced on stream sst)
ncclResult_t copy(void* dst, hipStream_t rst, const void* src, hipStream_t sst, size_t bytes) {
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    hipError_t e = hipEventRecord(ev, sst);
    if (e == hipSuccess) e = hipStreamWaitEvent(rst, ev, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, rst);
    if (e == hipSuccess) e = hipEventRecord(ev, rst);           // the sender's stream may not run ahead of the copy
    if (e == hipSuccess) e = hipStreamWaitEvent(sst, ev, 0);
    (void)hipEventDestroy(ev);
    return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t flush() {
    ncclResult_t rc = ncclSuccess;
    for (const Op& r : ops) {
        if (r.kind == 1) {
            const Op* s = nullptr;
            for (const Op& q : ops)
                if (q.kind == 0 && q.c->g == r.c->g && q.c->rank == r.peer && q.peer == r.c->rank && q.bytes == r.bytes) { s = &q; break; }
            if (!s) { rc = ncclInvalidUsage; continue; }
            const ncclResult_t e = copy(r.rbuf, r.st, s->sbuf, s->st, r.bytes);
            if (e != ncclSuccess) rc = e;
        } else if (r.kind == 2 && r.c->shm) {
            const ncclResult_t e = shm_allgather(r.c, r.sbuf, r.rbuf, r.bytes, r.st);
            if (e != ncclSuccess) rc = e;
        } else if (r.kind == 2) {
            int found = 0;
            for (const Op& q : ops)
                if (q.kind == 2 && q.c->g == r.c->g && q.bytes == r.bytes) {
                    const ncclResult_t e = copy((char*)r.rbuf + (size_t)q.c->rank * r.bytes, r.st, q.sbuf, q.st, r.bytes);
                    if (e != ncclSuccess) rc = e;
                    ++found;
                }
            if (found != r.c->n) rc = ncclInvalidUsage;          // a rank of this process did not take part
        }
    }
    ops.clear();
    return rc;
}

ncclResult_t push(const Op& o) {
    std::lock_guard<std::mutex> l(mu);
    ops.push_back(o);
    return depth > 0 ? ncclSuccess : flush();
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "nbls_loop_%d_%lld", (int)getpid(), (long long)(now_s() * 1e6));
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm* c = new Comm{new Group{nranks}, rank, nranks};
    if (nranks > 1) {                                // ranks are processes: map the file named by the id
        id.internal[sizeof(id.internal) - 1] = 0;
        char path[256];
        snprintf(path, sizeof(path), "/dev/shm/%s", id.internal);
        const size_t total = sizeof(Shm) + (size_t)MAXR * SLOT;
        const int fd = open(path, O_RDWR | O_CREAT, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)total) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
        void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);     // (a new file reads as zeros: counters start at 0)
        close(fd);
        if (p == MAP_FAILED) { delete c; return ncclSystemError; }
        c->shm = (Shm*)p;
        // everyone has mapped it before rank 0 removes the name (the memory lives on until the last unmap)
        c->shm->arrive[rank].store(-1, std::memory_order_release);
        bool ok = true;
        const double t0 = now_s();
        for (int q = 0; q < nranks && ok; ++q)
            while (c->shm->arrive[q].load(std::memory_order_acquire) != -1) { if (now_s() - t0 > 120.0) { ok = false; break; } usleep(50); }
        if (rank == 0) unlink(path);
        if (!ok) { delete c; return ncclSystemError; }
    }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int*) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    Group* g = new Group{ndev};
    for (int i = 0; i < ndev; ++i) comms[i] = (ncclComm_t) new Comm{g, i, ndev};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete (Comm*)c; return ncclSuccess; }     // (the Group leaks: test process)
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> l(mu); ++depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    std::lock_guard<std::mutex> l(mu);
    if (depth <= 0) return ncclInvalidUsage;
    return --depth == 0 ? flush() : ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    return push(Op{0, buf, nullptr, count * tsize(t), peer, (Comm*)c, st});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    return push(Op{1, nullptr, buf, count * tsize(t), peer, (Comm*)c, st});
}
ncclResult_t ncclAllGather(const void* sbuf, void* rbuf, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
    return push(Op{2, sbuf, rbuf, count * tsize(t), -1, (Comm*)c, st});
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "loopback stand-in: invalid use or HIP error"; }
}
