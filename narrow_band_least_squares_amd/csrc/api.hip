// C ABI of libnbls_hip.so (see include/nbls.h): handle, HBM buffers, plan, launch order.
#include "nbls_internal.h"
#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>

namespace {

std::mutex g_err_mu;
std::string g_err;   // error of a failed nbls_create

int fail(nbls_handle* h, int code, const std::string& msg) {
    if (h) { std::lock_guard<std::mutex> l(h->err_mu); h->err = msg; }
    else { std::lock_guard<std::mutex> l(g_err_mu); g_err = msg; }
    return code;
}

#define HIPCHK(h, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, e_ == hipErrorOutOfMemory ? NBLS_ERR_NOMEM : NBLS_ERR_HIP,          \
                        std::string(#call) + ": " + hipGetErrorString(e_));                    \
    } while (0)

// Blocking copy ON THE HANDLE'S STREAM.  The streams are non-blocking (hipStreamNonBlocking) and nothing in
// the library touches the NULL stream: a plain hipMemcpy would wait for every other handle of the process
// that is busy on this GPU (the band groups of a pipelined call run on several handles at once).
hipError_t copy_sync(nbls_handle* h, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, h->stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) h->work_queued = false;       // (the solve stream is joined into this one at the end of a pass)
    return e;
}

template <typename T>
int ensure(nbls_handle* h, T** p, size_t* cap, size_t need_bytes) {
    if (*p && *cap >= need_bytes) return 0;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    if (need_bytes == 0) need_bytes = 8;
    hipError_t e = hipMalloc((void**)p, need_bytes);
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(h, NBLS_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(need_bytes) + "): " + hipGetErrorString(e));
    }
    *cap = need_bytes;
    return 0;
}

// Upload a small plan table.  The allocation is kept and reused while it is big enough (hipFree
// synchronises the device and hipMalloc is slow: a plan uploads ~15 of these); its capacity is
// remembered in h->caps under the address of the pointer member.
template <typename T>
int alloc_copy(nbls_handle* h, T** p, const T* src, size_t n) {
    const size_t need = (n ? n : 1) * sizeof(T);
    const bool owned = h->arena_owned.count((const void*)p) != 0;
    {
        const size_t slot = (need + 63) & ~(size_t)63;
        if (h->arena_mode && h->stage && h->d_parena && h->stage_used + slot <= h->stage_cap) {
            // a table of nbls_plan: a place in the arena, no copy of its own (StreamGuard sends the arena in one piece)
            if (*p && !owned) { (void)hipFree(*p); h->caps.erase((const void*)p); }
            if (n) memcpy(h->stage + h->stage_used, src, n * sizeof(T));
            *p = (T*)(h->d_parena + h->stage_used);
            h->stage_used += slot;
            h->arena_owned.insert((const void*)p);
            return 0;
        }
    }
    if (owned) { *p = nullptr; h->arena_owned.erase((const void*)p); h->caps[(const void*)p] = 0; }   // back to an allocation of its own
    size_t& cap = h->caps[(const void*)p];
    if (!*p || cap < need) {
        if (*p) { (void)hipFree(*p); *p = nullptr; cap = 0; }
        HIPCHK(h, hipMalloc((void**)p, need));
        cap = need;
    }
    // queued on the handle's UPLOAD stream (highest priority: a plan made while other handles' passes fill the GPU
    // must not wait behind them — on a low-priority compute stream the ~15 small copies of a plan took 5 ms instead
    // of 0.3); nbls_plan / nbls_set_geometry wait for that stream before they return (StreamGuard), i.e. before the
    // host buffers go away and before any kernel that reads the tables can be launched
    if (n) {
        const size_t bytes = n * sizeof(T), slot = (bytes + 63) & ~(size_t)63;
        const void* from = src;
        if (h->stage && h->stage_used + slot <= h->stage_cap) {      // through the pinned arena (reset by StreamGuard)
            memcpy(h->stage + h->stage_used, src, bytes);
            from = h->stage + h->stage_used;
            h->stage_used += slot;
        } else {
            h->stage_bypass = true;                                  // the caller's buffer is read: the guard has to wait
        }
        HIPCHK(h, hipMemcpyAsync(*p, from, bytes, hipMemcpyHostToDevice, h->up));
    }
    return 0;
}

// Host-side linear algebra of the DF2T cascade (state order [s1_0, s2_0, s1_1, s2_1, ...]), in long
// double: one step is s' = A s + g x.  Returns, for a chunk of C samples and carry groups of G chunks,
//   fw[t][d]    = (A^(C-1-t) g)[d]         zero-state end state  e = sum_t fw[t] x_t
//   mpow[j]     = (A^C)^j, j = 0..G        chunk / group transitions
// Waits for the handle's upload stream when the enclosing API call returns, on every exit path: host tables that
// alloc_copy queued must have been read by then.
// On entry the upload stream is made to wait (GPU-side, on this handle's own streams only: other handles' passes
// are not touched) for whatever pass of THIS handle is still queued: nbls_execute is asynchronous, and the tables a
// plan overwrites (d_sos, d_fw, d_M, d_unit_*, d_W, d_starts, d_xs, ...) may be under running kernels.
struct StreamGuard {
    nbls_handle* h;
    explicit StreamGuard(nbls_handle* hh) : h(hh) {
        // the staging arena of alloc_copy: everything queued from it has been read when the previous guard left
        if (!h->stage) {
            constexpr size_t kStage = (size_t)8 << 20;
            if (hipHostMalloc((void**)&h->stage, kStage, hipHostMallocDefault) == hipSuccess) h->stage_cap = kStage;
            else { h->stage = nullptr; h->stage_cap = 0; (void)hipGetLastError(); }
        }
        if (h->up_pending) { (void)hipStreamSynchronize(h->up); h->up_pending = false; }     // the arena is free again
        h->stage_used = 0;
        h->stage_bypass = false;
        if (h->stage && !h->d_parena && hipMalloc((void**)&h->d_parena, h->stage_cap) != hipSuccess) { h->d_parena = nullptr; (void)hipGetLastError(); }
        if (!h->ev_up && hipEventCreateWithFlags(&h->ev_up, hipEventDisableTiming) != hipSuccess) { h->ev_up = nullptr; (void)hipGetLastError(); }
        if (h->work_queued && h->up != h->stream && h->ev_plan) {
            for (hipStream_t s : {h->stream, h->stream2}) {
                if (!s) continue;
                if (hipEventRecord(h->ev_plan, s) == hipSuccess) (void)hipStreamWaitEvent(h->up, h->ev_plan, 0);
            }
        }
    }
    ~StreamGuard() {
        h->arena_mode = false;                           // (a plan that failed half-way: its tables are never used)
        // everything came from the arena: nothing of the caller's is still being read, and the kernels that need the
        // tables are ordered behind ev_up on the GPU (wait_uploads) — the host does not wait for the copies
        if (!h->stage_bypass && h->stage && h->ev_up && hipEventRecord(h->ev_up, h->up) == hipSuccess) { h->up_pending = true; return; }
        (void)hipStreamSynchronize(h->up);
    }
};

// GPU-side: the handle's compute streams wait for the plan / geometry tables still on their way (see ev_up).
void wait_uploads(nbls_handle* h) {
    if (!h->up_pending || !h->ev_up) return;
    for (hipStream_t s : {h->stream, h->stream2})
        if (s && s != h->up) (void)hipStreamWaitEvent(s, h->ev_up, 0);
}

// A few host threads kept asleep between plans (filter tables: long-double arithmetic, 30-100 us per band, on the critical
// path of every call — the GPU has nothing to do until the plan is through).  Starting a thread costs ~15 us, one after
// the other on the planning thread: eleven of them were a third of the 0.5 ms the tables of 48 bands took.  One job at a
// time; a second planner (the launch threads of a multi-GPU call plan side by side) computes its bands inline instead of
// waiting.  The pool is never destroyed (threads asleep at exit are simply gone with the process) and is rebuilt in a
// forked child, which inherits none of them.
struct TablePool {
    std::mutex m;
    std::condition_variable go, done;
    std::mutex busy;                       // held for the length of one job
    const std::function<void(int)>* job = nullptr;
    int njob = 0, left = 0;
    unsigned long gen = 0;
    int nthreads = 0;
    pid_t pid = 0;
    void worker(int t) {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)>* f;
            {
                std::unique_lock<std::mutex> l(m);
                go.wait(l, [&] { return gen != seen; });
                seen = gen;
                if (t >= njob) continue;
                f = job;
            }
            (*f)(t);
            std::lock_guard<std::mutex> l(m);
            if (--left == 0) done.notify_one();
        }
    }
};
TablePool* g_pool = nullptr;
std::mutex g_pool_mu;

// f(0) .. f(n-1), f(0) on the calling thread; returns false when the pool is taken (the caller then runs everything itself)
bool pool_run(int n, const std::function<void(int)>& f) {
    constexpr int kThreads = 12;
    TablePool* p;
    {
        std::lock_guard<std::mutex> l(g_pool_mu);
        if (!g_pool || g_pool->pid != getpid()) {
            p = new TablePool;                       // (an inherited pool of the parent process is left alone)
            p->pid = getpid();
            p->nthreads = kThreads;
            for (int t = 1; t < kThreads; ++t) std::thread(&TablePool::worker, p, t).detach();
            g_pool = p;
        }
        p = g_pool;
    }
    if (n > p->nthreads) return false;
    std::unique_lock<std::mutex> use(p->busy, std::try_to_lock);
    if (!use.owns_lock()) return false;
    if (n > 1) {
        std::lock_guard<std::mutex> l(p->m);
        p->job = &f;
        p->njob = n;
        p->left = n - 1;
        ++p->gen;
        p->go.notify_all();
    }
    f(0);
    if (n > 1) {
        std::unique_lock<std::mutex> l(p->m);
        p->done.wait(l, [&] { return p->left == 0; });
        p->job = nullptr;
    }
    return true;
}

void filter_tables(const double* sos, int S, int C, int G, double* fw, double* mpow) {
    constexpr int DM = 2 * NBLS_MAX_SECTIONS;
    const int D = 2 * S;
    typedef long double ld;
    // fixed-size stack matrices (D <= 16): this runs for every band of every plan
    auto step = [&](ld* st, ld x) {
        ld v = x;
        for (int s = 0; s < S; ++s) {
            const ld b0 = sos[s * 6 + 0], b1 = sos[s * 6 + 1], b2 = sos[s * 6 + 2];
            const ld a1 = sos[s * 6 + 4], a2 = sos[s * 6 + 5];
            const ld y = b0 * v + st[2 * s];
            const ld n1 = (b1 * v - a1 * y) + st[2 * s + 1];
            const ld n2 = b2 * v - a2 * y;
            st[2 * s] = n1;
            st[2 * s + 1] = n2;
            v = y;
        }
    };
    ld A[DM * DM], g[DM];
    for (int col = 0; col < D; ++col) {
        ld st[DM];
        for (int r = 0; r < D; ++r) st[r] = 0.0L;
        st[col] = 1.0L;
        step(st, 0.0L);
        for (int r = 0; r < D; ++r) A[r * D + col] = st[r];
    }
    for (int r = 0; r < D; ++r) g[r] = 0.0L;
    step(g, 1.0L);
    auto matmul = [&](const ld* X, const ld* Y, ld* o) {          // o = X Y (o distinct from X, Y)
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                ld acc = 0.0L;
                for (int k = 0; k < D; ++k) acc += X[i * D + k] * Y[k * D + j];
                o[i * D + j] = acc;
            }
    };
    // v = A^k g is the weight of sample t = C-1-k.  One DF2T step with input 0 IS the product A v
    // (A's columns were made by that same step), at O(S) instead of O(D^2) operations.
    ld v[DM];
    for (int d = 0; d < D; ++d) v[d] = g[d];
    for (int k = 0; k < C; ++k) {
        for (int d = 0; d < D; ++d) fw[(size_t)(C - 1 - k) * D + d] = (double)v[d];
        ld nv[DM];
        for (int i = 0; i < D; ++i) {
            ld acc = 0.0L;
            for (int q = 0; q < D; ++q) acc += A[i * D + q] * v[q];
            nv[i] = acc;
        }
        for (int d = 0; d < D; ++d) v[d] = nv[d];
    }
    ld M[DM * DM], Bm[DM * DM], T[DM * DM];                       // M = A^C by squaring
    for (int i = 0; i < D * D; ++i) { M[i] = 0.0L; Bm[i] = A[i]; }
    for (int i = 0; i < D; ++i) M[i * D + i] = 1.0L;
    for (int e = C; e > 0; e >>= 1) {
        if (e & 1) { matmul(M, Bm, T); for (int i = 0; i < D * D; ++i) M[i] = T[i]; }
        matmul(Bm, Bm, T);
        for (int i = 0; i < D * D; ++i) Bm[i] = T[i];
    }
    ld Pw[DM * DM];
    for (int i = 0; i < D * D; ++i) Pw[i] = 0.0L;
    for (int i = 0; i < D; ++i) Pw[i * D + i] = 1.0L;
    for (int j = 0; j <= G; ++j) {
        for (int i = 0; i < D * D; ++i) mpow[(size_t)j * D * D + i] = (double)Pw[i];
        matmul(Pw, M, T);
        for (int i = 0; i < D * D; ++i) Pw[i] = T[i];
    }
}

}  // namespace

extern "C" {

int nbls_version(void) { return 200; }

const char* nbls_last_error(const nbls_handle* h) {
    if (h) return h->err.c_str();
    std::lock_guard<std::mutex> l(g_err_mu);
    return g_err.c_str();
}

int nbls_device_count(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return 0;
    return ndev;
}

int nbls_create(int device_id, nbls_handle** out) {
    if (!out) return fail(nullptr, NBLS_ERR_ARG, "nbls_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NBLS_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev)
        return fail(nullptr, NBLS_ERR_ARG, "nbls_create: device_id out of range");
    nbls_handle* h = new nbls_handle();
    h->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete h;
        return fail(nullptr, NBLS_ERR_HIP, std::string("device init: ") + hipGetErrorString(e));
    }
    (void)hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking);
    {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) greatest = 0;
        if (hipStreamCreateWithPriority(&h->up, hipStreamNonBlocking, greatest) != hipSuccess) {
            h->up = h->stream;             // no priorities on this device: uploads share the compute stream
            (void)hipGetLastError();
        }
    }
    if (hipDeviceGetAttribute(&h->num_cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess) h->num_cus = 0;
    for (int i = 0; i < 4; ++i) (void)hipEventCreate(&h->ev[i]);
    if (hipEventCreateWithFlags(&h->ev_xd, hipEventDisableTiming) != hipSuccess) { h->ev_xd = nullptr; (void)hipGetLastError(); }
    if (hipEventCreateWithFlags(&h->ev_plan, hipEventDisableTiming) != hipSuccess) { h->ev_plan = nullptr; (void)hipGetLastError(); }
    *out = h;
    return NBLS_OK;
}

void nbls_destroy(nbls_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->up) (void)hipStreamSynchronize(h->up);
    (void)nbls_comm_destroy(h);
    void* bufs[] = {h->d_trace, h->d_xij, h->d_pair, h->d_xpinv, h->d_sos, h->d_M, h->d_tl, h->d_tr,
                    h->d_W, h->d_inc, h->d_nwin, h->d_unit_off, h->d_unit_band, h->d_unit_win, h->d_filt, h->d_cstate, h->d_cstate2, h->d_tstate,
                    h->d_lag, h->d_cmax, h->d_res /* vel, baz, mdccm, sigma_tau, mask */, h->d_z, h->d_wts,
                    h->d_unc, h->d_starts, h->d_rew, h->d_xs, h->d_xc, h->d_xss, h->d_qbuf, h->d_qmeta, h->d_cand, h->d_fw, h->d_gend, h->d_gin, h->d_win_off, h->d_stamps, h->d_seg_state};
    {
        std::unordered_set<const void*> in_arena;       // members that point into d_parena
        for (const void* m : h->arena_owned) in_arena.insert(*(void* const*)m);
        for (void* b : bufs) if (b && !in_arena.count(b)) (void)hipFree(b);
        if (h->d_parena) (void)hipFree(h->d_parena);
    }
    for (int i = 0; i < 4; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    for (hipEvent_t e : h->bev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->pev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->rev) (void)hipEventDestroy(e);
    if (h->cstream) { (void)hipStreamSynchronize(h->cstream); (void)hipStreamDestroy(h->cstream); }
    if (h->ustream) { (void)hipStreamSynchronize(h->ustream); (void)hipStreamDestroy(h->ustream); }
    for (hipEvent_t e : h->uev) (void)hipEventDestroy(e);
    if (h->ev_uprev) (void)hipEventDestroy(h->ev_uprev);
    if (h->h_res) (void)hipHostFree(h->h_res);
    if (h->ev_xd) (void)hipEventDestroy(h->ev_xd);
    if (h->ev_plan) (void)hipEventDestroy(h->ev_plan);
    if (h->ev_up) (void)hipEventDestroy(h->ev_up);
    if (h->up && h->up != h->stream) (void)hipStreamDestroy(h->up);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->stage) (void)hipHostFree(h->stage);
    delete h;
}

// Declare the trace: allocation and shape, no samples yet (h->trace_loaded = false).
static int trace_shape_impl(nbls_handle* h, int32_t nchans, int64_t npts, double fs) {
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t pad = (npts + 63) / 64 * 64;
    const size_t need = (size_t)nchans * pad * sizeof(double);
    if (!h->d_trace || h->cap_trace < need) {
        if (h->d_trace) { (void)hipFree(h->d_trace); h->d_trace = nullptr; h->cap_trace = 0; }
        HIPCHK(h, hipMalloc((void**)&h->d_trace, need));
        h->cap_trace = need;
    }
    if (h->nchans != nchans && h->d_xij) {      // a geometry of another array size is stale
        (void)hipFree(h->d_xij); h->d_xij = nullptr;
        h->caps.erase((const void*)&h->d_xij);
        h->npairs = 0;
    }
    h->nchans = nchans;
    h->npts = npts;
    h->npts_pad = pad;
    h->fs = fs;
    h->planned = false;
    h->trace_loaded = false;
    h->rows_landed.store(0, std::memory_order_release);
    h->upload_state.store(2, std::memory_order_release);
    return NBLS_OK;
}

// Copy the samples of a declared trace, row by row on the handle's upload stream, an event behind every row (a pass
// queued meanwhile on another thread filters the rows as they land, see wait_rows).  Touches only d_trace, the upload
// stream and (on failure) the error text: nbls_upload_rows may therefore run beside nbls_set_geometry / nbls_plan /
// nbls_execute of the same handle on another thread.
static int trace_copy_impl(nbls_handle* h, const double* const* rows, const double* flat) {
    struct Guard {
        nbls_handle* h; bool ok = false;
        ~Guard() { h->upload_state.store(ok ? 0 : -1, std::memory_order_release); }
    } guard{h};
    h->rows_landed.store(0, std::memory_order_release);
    h->upload_state.store(1, std::memory_order_release);
    HIPCHK(h, hipSetDevice(h->device));
    const int32_t nchans = h->nchans;
    const int64_t npts = h->npts, pad = h->npts_pad;
    if (!h->ustream) HIPCHK(h, hipStreamCreateWithFlags(&h->ustream, hipStreamNonBlocking));
    if (!h->ev_uprev) HIPCHK(h, hipEventCreateWithFlags(&h->ev_uprev, hipEventDisableTiming));
    while ((int)h->uev.size() < nchans) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->uev.push_back(e);
    }
    // behind whatever is still queued on the compute streams: an earlier pass may be reading the trace this one replaces
    for (hipStream_t s : {h->stream, h->stream2}) {
        if (!s) continue;
        HIPCHK(h, hipEventRecord(h->ev_uprev, s));
        HIPCHK(h, hipStreamWaitEvent(h->ustream, h->ev_uprev, 0));
    }
    if (pad > npts)
        HIPCHK(h, hipMemset2DAsync(h->d_trace + npts, pad * sizeof(double), 0, (pad - npts) * sizeof(double), nchans, h->ustream));
    if (flat) {
        HIPCHK(h, hipMemcpy2DAsync(h->d_trace, pad * sizeof(double), flat, npts * sizeof(double),
                                   npts * sizeof(double), nchans, hipMemcpyHostToDevice, h->ustream));
        for (int c = 0; c < nchans; ++c) HIPCHK(h, hipEventRecord(h->uev[c], h->ustream));
        h->rows_landed.store(nchans, std::memory_order_release);
    } else {
        for (int c = 0; c < nchans; ++c) {
            HIPCHK(h, hipMemcpyAsync(h->d_trace + (size_t)c * pad, rows[c], npts * sizeof(double), hipMemcpyHostToDevice, h->ustream));
            HIPCHK(h, hipEventRecord(h->uev[c], h->ustream));
            h->rows_landed.store(c + 1, std::memory_order_release);
        }
    }
    // the caller's buffers may be reused as soon as this returns (pageable sources are staged before
    // hipMemcpyAsync returns; pinned ones are still being read): wait for the copies
    HIPCHK(h, hipStreamSynchronize(h->ustream));
    h->trace_loaded = true;
    guard.ok = true;
    return NBLS_OK;
}

// Host side of the row pipeline: wait until more than `done` channels of the trace are in flight behind their events
// (-> how many), the upload has failed (-> -1), nothing arrives (-> -2) or the shape was declared but no upload announced
// (nbls_expect_upload) or started (-> -3).  With no upload under way every channel is there.
static int wait_rows(nbls_handle* h, int done) {
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        const int st = h->upload_state.load(std::memory_order_acquire);
        const int landed = h->rows_landed.load(std::memory_order_acquire);
        if (st < 0) return -1;
        if (st == 0) return h->trace_loaded ? h->nchans : -1;
        if (st == 2) return -3;                  // declared, and nobody has announced the samples
        if (landed > done) return landed;
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return -2;
        std::this_thread::yield();
    }
}

static int set_trace_impl(nbls_handle* h, const double* const* rows, const double* flat, int32_t nchans, int64_t npts,
                          double fs) {
    const int rc = trace_shape_impl(h, nchans, npts, fs);
    return rc ? rc : trace_copy_impl(h, rows, flat);
}

int nbls_set_trace_shape(nbls_handle* h, int32_t nchans, int64_t npts, double fs) {
    if (!h) return NBLS_ERR_ARG;
    if (nchans < 1 || npts < 1 || !(fs > 0.0)) return fail(h, NBLS_ERR_ARG, "nbls_set_trace_shape: bad argument");
    return trace_shape_impl(h, nchans, npts, fs);
}

int nbls_upload_rows(nbls_handle* h, const double* const* rows, int32_t nchans, int64_t npts) {
    if (!h) return NBLS_ERR_ARG;
    if (!rows || nchans != h->nchans || npts != h->npts || !h->d_trace)
        return fail(h, NBLS_ERR_ARG, "nbls_upload_rows: rows do not match the shape declared with nbls_set_trace_shape");
    for (int c = 0; c < nchans; ++c)
        if (!rows[c]) return fail(h, NBLS_ERR_ARG, "nbls_upload_rows: null row");
    return trace_copy_impl(h, rows, nullptr);
}

int nbls_expect_upload(nbls_handle* h) {
    if (!h) return NBLS_ERR_ARG;
    int declared = 2;
    if (!h->upload_state.compare_exchange_strong(declared, 3, std::memory_order_acq_rel))
        return fail(h, NBLS_ERR_STATE, "nbls_expect_upload: no trace shape declared (nbls_set_trace_shape) or its samples are there already");
    return NBLS_OK;
}

int nbls_abort_upload(nbls_handle* h) {
    if (!h) return NBLS_ERR_ARG;
    int st = h->upload_state.load(std::memory_order_acquire);
    while ((st == 2 || st == 3) && !h->upload_state.compare_exchange_weak(st, -1, std::memory_order_acq_rel)) {}
    return NBLS_OK;
}

int nbls_set_trace(nbls_handle* h, const double* trace, int32_t nchans, int64_t npts, double fs) {
    if (!h) return NBLS_ERR_ARG;
    if (!trace || nchans < 1 || npts < 1 || !(fs > 0.0)) return fail(h, NBLS_ERR_ARG, "nbls_set_trace: bad argument");
    return set_trace_impl(h, nullptr, trace, nchans, npts, fs);
}

int nbls_set_trace_from(nbls_handle* h, const nbls_handle* src) {
    if (!h || !src) return NBLS_ERR_ARG;
    if (!src->d_trace) return fail(h, NBLS_ERR_STATE, "nbls_set_trace_from: the source handle has no trace");
    if (h == src) return NBLS_OK;
    if (h->device != src->device) return fail(h, NBLS_ERR_ARG, "nbls_set_trace_from: handles are on different devices");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t need = (size_t)src->nchans * src->npts_pad * sizeof(double);
    if (!h->d_trace || h->cap_trace < need) {
        if (h->d_trace) { (void)hipFree(h->d_trace); h->d_trace = nullptr; h->cap_trace = 0; }
        HIPCHK(h, hipMalloc((void**)&h->d_trace, need));
        h->cap_trace = need;
    }
    // the source's upload has completed (nbls_set_trace* return after it) or is still running on another thread
    // (nbls_upload_rows): wait for its last row; the copy is ordered on THIS handle's stream
    {
        nbls_handle* s_ = const_cast<nbls_handle*>(src);
        int landed = 0;
        while (landed >= 0 && landed < src->nchans) landed = wait_rows(s_, landed);
        if (landed < 0) return fail(h, NBLS_ERR_STATE, "nbls_set_trace_from: the source handle's trace did not arrive");
        if (!src->uev.empty() && (int)src->uev.size() >= src->nchans) HIPCHK(h, hipStreamWaitEvent(h->stream, src->uev[src->nchans - 1], 0));
    }
    HIPCHK(h, hipMemcpyAsync(h->d_trace, src->d_trace, need, hipMemcpyDeviceToDevice, h->stream));
    if (h->nchans != src->nchans && h->d_xij) {
        (void)hipFree(h->d_xij); h->d_xij = nullptr;
        h->caps.erase((const void*)&h->d_xij);
        h->npairs = 0;
    }
    h->nchans = src->nchans;
    h->npts = src->npts;
    h->npts_pad = src->npts_pad;
    h->fs = src->fs;
    h->planned = false;
    h->trace_loaded = true;               // (ordered on this handle's stream before its next pass)
    h->rows_landed.store(h->nchans, std::memory_order_release);
    h->upload_state.store(0, std::memory_order_release);
    return NBLS_OK;
}

int nbls_set_trace_rows(nbls_handle* h, const double* const* rows, int32_t nchans, int64_t npts, double fs) {
    if (!h) return NBLS_ERR_ARG;
    if (!rows || nchans < 1 || npts < 1 || !(fs > 0.0)) return fail(h, NBLS_ERR_ARG, "nbls_set_trace_rows: bad argument");
    for (int c = 0; c < nchans; ++c)
        if (!rows[c]) return fail(h, NBLS_ERR_ARG, "nbls_set_trace_rows: NULL row pointer");
    return set_trace_impl(h, rows, nullptr, nchans, npts, fs);
}

int nbls_set_geometry(nbls_handle* h, const double* xij, const int32_t* pair_idx, const double* xpinv,
                      int32_t npairs) {
    if (!h) return NBLS_ERR_ARG;
    if (!xij || !pair_idx || !xpinv) return fail(h, NBLS_ERR_ARG, "nbls_set_geometry: NULL pointer");
    if (npairs < 3) return fail(h, NBLS_ERR_GEOMETRY, "need at least 3 array elements (3 pairs)");
    if (npairs > NBLS_MAX_PAIRS) return fail(h, NBLS_ERR_UNSUPPORTED, "more than 512 pairs (32 elements) not supported");
    // rank check of the co-array (2 unknowns)
    double sxx = 0, sxy = 0, syy = 0;
    for (int k = 0; k < npairs; ++k) { sxx += xij[2*k]*xij[2*k]; sxy += xij[2*k]*xij[2*k+1]; syy += xij[2*k+1]*xij[2*k+1]; }
    const double det = sxx * syy - sxy * sxy;
    if (!(det > 1e-12 * (sxx + syy) * (sxx + syy)))
        return fail(h, NBLS_ERR_GEOMETRY, "co-array is rank deficient (collinear array)");
    // the same geometry as the handle already holds (every band group of every call of one array): nothing to upload
    const size_t n2 = (size_t)npairs * 2;
    if (h->d_xij && h->d_pair && h->d_xpinv && h->npairs == npairs && h->h_xij.size() == n2 && h->h_pair.size() == n2 &&
        h->h_xpinv.size() == n2 && memcmp(h->h_xij.data(), xij, n2 * sizeof(double)) == 0 &&
        memcmp(h->h_pair.data(), pair_idx, n2 * sizeof(int32_t)) == 0 && memcmp(h->h_xpinv.data(), xpinv, n2 * sizeof(double)) == 0) {
        h->planned = false;
        return NBLS_OK;
    }
    HIPCHK(h, hipSetDevice(h->device));
    StreamGuard guard(h);
    int rc;
    h->h_xij.clear();                        // (a failed upload leaves no stale "already there" record)
    if ((rc = alloc_copy(h, &h->d_xij, xij, n2))) return rc;
    if ((rc = alloc_copy(h, &h->d_pair, pair_idx, n2))) return rc;
    if ((rc = alloc_copy(h, &h->d_xpinv, xpinv, n2))) return rc;
    h->h_xij.assign(xij, xij + n2);
    h->h_pair.assign(pair_idx, pair_idx + n2);
    h->h_xpinv.assign(xpinv, xpinv + n2);
    h->npairs = npairs;
    h->planned = false;
    return NBLS_OK;
}

int nbls_set_window_ranges(nbls_handle* h, int32_t nbands, const int32_t* first, const int32_t* count) {
    if (!h) return NBLS_ERR_ARG;
    // consumed by the NEXT nbls_plan; an existing plan keeps the ranges it was made with
    if (nbands <= 0 || !first || !count) {       // reset: every band processes all its windows
        h->win_first.clear();
        h->win_count.clear();
        return NBLS_OK;
    }
    for (int b = 0; b < nbands; ++b)
        if (first[b] < 0) return fail(h, NBLS_ERR_ARG, "nbls_set_window_ranges: negative first window");
    h->win_first.assign(first, first + nbands);
    h->win_count.assign(count, count + nbands);
    return NBLS_OK;
}

int nbls_plan(nbls_handle* h, int32_t nbands, const double* sos, int32_t nsections, int32_t zero_phase,
              const double* taper_left, const double* taper_right, int32_t taper_len,
              const int32_t* winlen, const int32_t* wininc, int32_t vector_len,
              const nbls_lts_params* lts, int32_t xcorr_impl) {
    if (!h) return NBLS_ERR_ARG;
    h->planned = false;
    if (!h->d_trace) return fail(h, NBLS_ERR_STATE, "nbls_plan: no trace set");
    if (nbands < 1 || (!sos && nsections != 0) || !winlen || !wininc || vector_len < 1)
        return fail(h, NBLS_ERR_ARG, "nbls_plan: bad argument");
    if (nsections < 0 || nsections > NBLS_MAX_SECTIONS)
        return fail(h, NBLS_ERR_UNSUPPORTED, "nbls_plan: 0..8 second-order sections supported");
    if (taper_len < 0 || 2 * (int64_t)taper_len > h->npts || (taper_len > 0 && (!taper_left || !taper_right)))
        return fail(h, NBLS_ERR_ARG, "nbls_plan: bad taper");
    // geometry is optional for a filter-only plan (filter_data()); execute checks it
    const int P = h->d_xij ? h->npairs : 1;
    if (h->d_xij && (int64_t)h->nchans * (h->nchans - 1) / 2 != P)
        return fail(h, NBLS_ERR_ARG, "nbls_plan: geometry pair count does not match the trace channel count");
    if (lts) {
        if (!h->d_xij) return fail(h, NBLS_ERR_STATE, "nbls_plan: LTS needs the geometry");
        if (h->nchans < 4) return fail(h, NBLS_ERR_GEOMETRY, "LTS needs at least 4 array elements");
        if (lts->nstarts < 1 || lts->nstarts > NBLS_MAX_STARTS || !lts->starts || !lts->rew_table)
            return fail(h, NBLS_ERR_ARG, "nbls_plan: bad LTS starts");
        if (lts->h < 2 || lts->h > P) return fail(h, NBLS_ERR_ARG, "nbls_plan: LTS h out of range");
        if (lts->ncand < 1 || lts->ncand > NBLS_MAX_CAND) return fail(h, NBLS_ERR_ARG, "nbls_plan: ncand out of range");
        for (int i = 0; i < lts->nstarts * 4; ++i)
            if (lts->starts[i] >= P) return fail(h, NBLS_ERR_ARG, "nbls_plan: start index out of range");
    }
    // The filter tables first: pure host arithmetic.  Every HIP call of the plan comes AFTER them — while the trace of a
    // pipelined call is still going up on another thread (nbls_upload_rows) each HIP call waits for the runtime's lock
    // behind a row copy (~0.15 ms apiece), and the first launch of the call waits for this plan.
    const auto tp0 = std::chrono::steady_clock::now();
    const int D = 2 * nsections;
    const int GG = NBLS_FILTER_GROUP;
    // (host tables kept in the handle between plans: a fresh 1.2 MB per plan is 300 page faults on the table threads, on the
    //  critical path of every call — and trimmed back to the OS when it is freed)
    std::vector<double>& M = h->hp_M;
    std::vector<double>& FW = h->hp_FW;
    M.resize((size_t)nbands * (GG + 1) * D * D);
    FW.resize((size_t)nbands * NBLS_FILTER_CHUNK * D);
    {
        // long-double table arithmetic, 30-100 us per band: the bands are dealt to a few host threads (the
        // plan sits on the critical path of a call: the GPU has nothing to do until it is through)
        double* const fwp = FW.data();
        double* const mp = M.data();
        int nt = nsections > 0 ? std::max(1, std::min(12, nbands / 2)) : 0;     // 2+ bands per thread of the pool (TablePool)
        const std::function<void(int)> work = [&](int t) {
            for (int b = t; b < nbands; b += nt)
                filter_tables(sos + (size_t)b * nsections * 6, nsections, NBLS_FILTER_CHUNK, GG,
                              fwp + (size_t)b * NBLS_FILTER_CHUNK * D, mp + (size_t)b * (GG + 1) * D * D);
        };
        if (nt > 0 && !pool_run(nt, work)) {       // the pool is busy with another handle's plan: all bands here
            nt = 1;
            work(0);
        }
    }
    const auto tp1 = std::chrono::steady_clock::now();

    HIPCHK(h, hipSetDevice(h->device));
    StreamGuard guard(h);
    h->planned = false;
    h->res_loaded = false;
    h->arena_mode = true;                                // alloc_copy: places in the arena, ONE upload at the end

    h->W.assign(winlen, winlen + nbands);
    h->inc.assign(wininc, wininc + nbands);
    h->nwin.resize(nbands);
    h->unit_off.resize(nbands + 1);
    std::vector<int32_t> woff(nbands, 0);
    int64_t U = 0;
    int maxW = 0, uniW = nbands > 0 ? winlen[0] : 0;
    for (int b = 0; b < nbands; ++b) {
        const int W = winlen[b], inc = wininc[b];
        if (W != uniW) uniW = 0;
        if (W < 2 || inc < 1)
            return fail(h, NBLS_ERR_ARG, "nbls_plan: window length must be at least 2 samples, the hop at least 1");
        // len(arange(0, npts - W, inc))
        const int64_t span = h->npts - W;
        int64_t n = span > 0 ? (span + inc - 1) / inc : 0;
        if (n > vector_len) return fail(h, NBLS_ERR_ARG, "nbls_plan: vector_len smaller than a band's window count");
        int64_t first = 0;
        if ((int)h->win_first.size() == nbands) {          // window sharding: this handle's slice of the band
            first = h->win_first[b] < n ? h->win_first[b] : n;
            int64_t cnt = h->win_count[b] < 0 ? n - first : h->win_count[b];
            if (cnt > n - first) cnt = n - first;
            n = cnt;
        }
        woff[b] = (int32_t)first;
        h->nwin[b] = (int32_t)n;
        h->unit_off[b] = (int32_t)U;
        U += n;
        if (W > maxW) maxW = W;
    }
    h->unit_off[nbands] = (int32_t)U;
    h->woff = woff;
    if (U > 0x7fffffffLL / (P > 0 ? P : 1)) return fail(h, NBLS_ERR_UNSUPPORTED, "nbls_plan: too many (unit, pair) items for one launch");
    h->nunits = U;
    h->maxW = maxW;
    h->uniW = uniW;
    h->nbands = nbands;
    h->nsections = nsections;
    h->zero_phase = zero_phase ? 1 : 0;
    h->taper_len = taper_len;
    h->vector_len = vector_len;
    h->xcorr_impl = xcorr_impl;
    h->nchunks = (h->npts + NBLS_FILTER_CHUNK - 1) / NBLS_FILTER_CHUNK;

    int rc;
    if ((rc = alloc_copy(h, &h->d_fw, FW.data(), FW.size()))) return rc;
    if ((rc = alloc_copy(h, &h->d_sos, sos, (size_t)nbands * nsections * 6))) return rc;
    if (nsections == 0 && nbands != 1)
        return fail(h, NBLS_ERR_ARG, "nbls_plan: an unfiltered plan has exactly one band");
    if ((rc = alloc_copy(h, &h->d_M, M.data(), M.size()))) return rc;
    {   // the ramps (1 % of the trace each: 0.14 MB at cfg-3, a quarter of a plan's upload) only when they differ from what is there
        const size_t tn = (size_t)taper_len;
        const bool same = h->d_tl && h->d_tr && h->h_tl.size() == tn && h->h_tr.size() == tn &&
                          (tn == 0 || (memcmp(h->h_tl.data(), taper_left, tn * sizeof(double)) == 0 &&
                                       memcmp(h->h_tr.data(), taper_right, tn * sizeof(double)) == 0));
        if (!same) {
            h->h_tl.clear();
            h->h_tr.clear();
            h->arena_mode = false;                       // allocations of their own: they outlive the plan
            rc = alloc_copy(h, &h->d_tl, taper_left, tn);
            if (!rc) rc = alloc_copy(h, &h->d_tr, taper_right, tn);
            h->arena_mode = true;
            if (rc) return rc;
            if (tn) { h->h_tl.assign(taper_left, taper_left + tn); h->h_tr.assign(taper_right, taper_right + tn); }
        }
    }
    if ((rc = alloc_copy(h, &h->d_W, h->W.data(), (size_t)nbands))) return rc;
    if ((rc = alloc_copy(h, &h->d_inc, h->inc.data(), (size_t)nbands))) return rc;
    if ((rc = alloc_copy(h, &h->d_nwin, h->nwin.data(), (size_t)nbands))) return rc;
    if ((rc = alloc_copy(h, &h->d_unit_off, h->unit_off.data(), (size_t)nbands + 1))) return rc;
    if ((rc = alloc_copy(h, &h->d_win_off, woff.data(), (size_t)nbands))) return rc;
    std::vector<int32_t>& ub = h->hp_ub;
    ub.resize((size_t)U);
    for (int b = 0; b < nbands; ++b)
        for (int64_t u = h->unit_off[b]; u < h->unit_off[b + 1]; ++u) ub[(size_t)u] = b;
    if ((rc = alloc_copy(h, &h->d_unit_band, ub.data(), (size_t)U))) return rc;
    {
        std::vector<int32_t>& uw = h->hp_uw;
        uw.resize((size_t)(U > 0 ? U : 1));
        for (int b = 0; b < nbands; ++b)
            for (int64_t u = h->unit_off[b]; u < h->unit_off[b + 1]; ++u) uw[(size_t)u] = (int32_t)(u - h->unit_off[b]) + woff[b];
        if ((rc = alloc_copy(h, &h->d_unit_win, uw.data(), (size_t)U))) return rc;
    }

    const auto tp2 = std::chrono::steady_clock::now();
    const size_t nseries = (size_t)nbands * h->nchans;
    // + 64 bytes: the verifier's 16-byte copies of a window that starts on an odd sample read one sample past its end
    if ((rc = ensure(h, &h->d_filt, &h->cap_filt, nseries * h->npts_pad * sizeof(double) + 64))) return rc;
    if ((rc = ensure(h, &h->d_cstate, &h->cap_cstate, nseries * h->nchunks * D * sizeof(double)))) return rc;
    if ((rc = ensure(h, &h->d_cstate2, &h->cap_cstate2, nseries * h->nchunks * D * sizeof(double)))) return rc;
    if (h->d_tstate && !(h->zero_phase && nsections > 0 && nsections <= 4)) {
        (void)hipFree(h->d_tstate); h->d_tstate = nullptr; h->cap_tstate = 0;     // this plan filters in the stored form
    }
    if (h->zero_phase && nsections > 0 && nsections <= 4) {
        // tile-boundary states of the recompute form (2S doubles per 16 samples: a quarter of the filtered buffer at
        // two sections; beyond four sections they would be as large as the samples they replace, and the stored form
        // is used).  Without room for them the filter writes and re-reads the forward output instead: same results.
        const size_t need = nseries * h->nchunks * (NBLS_FILTER_CHUNK / NBLS_FILTER_TILE) * D * sizeof(double);
        if (need > h->cap_tstate) {
            if (h->d_tstate) { (void)hipFree(h->d_tstate); h->d_tstate = nullptr; h->cap_tstate = 0; }
            if (hipMalloc((void**)&h->d_tstate, need) == hipSuccess) h->cap_tstate = need;
            else { h->d_tstate = nullptr; (void)hipGetLastError(); }
        }
    }
    const size_t ngroups = (size_t)((h->nchunks + NBLS_FILTER_GROUP - 1) / NBLS_FILTER_GROUP);
    if ((rc = ensure(h, &h->d_gend, &h->cap_gend, nseries * ngroups * D * sizeof(double)))) return rc;
    if ((rc = ensure(h, &h->d_gin, &h->cap_gin, nseries * ngroups * D * sizeof(double)))) return rc;
    // results: every buffer has its own capacity; the views into the result block are recomputed by
    // every plan (a smaller plan after a bigger one keeps the allocation but must move the grid offsets)
    const size_t cells = (size_t)nbands * vector_len;
    h->mask_bytes = (P + 7) / 8;
    h->res_bytes = cells * (4 * sizeof(double) + (size_t)h->mask_bytes);
    h->d_vel = h->d_baz = h->d_mdccm = h->d_sig = nullptr;
    h->d_mask = nullptr;
    if ((rc = ensure(h, &h->d_res, &h->cap_res, h->res_bytes > h->reserve_res ? h->res_bytes : h->reserve_res))) return rc;
    if ((rc = ensure(h, &h->d_lag, &h->cap_lag, cells * P * sizeof(int32_t)))) return rc;
    if ((rc = ensure(h, &h->d_cmax, &h->cap_cmax, cells * P * sizeof(double)))) return rc;
    if ((rc = ensure(h, &h->d_z, &h->cap_z, 2 * cells * sizeof(double)))) return rc;
    if ((rc = ensure(h, &h->d_wts, &h->cap_wts, cells * P))) return rc;
    if (h->want_unc && (rc = ensure(h, &h->d_unc, &h->cap_unc, 2 * cells * sizeof(double)))) return rc;
    h->d_vel = (double*)h->d_res;
    h->d_baz = h->d_vel + cells;
    h->d_mdccm = h->d_baz + cells;
    h->d_sig = h->d_mdccm + cells;
    h->d_mask = h->d_res + 4 * cells * sizeof(double);

    {   // window groups (consecutive bands of one window length) and, per group, whether the int8 screening correlator
        // applies; "auto" = screening wherever it applies, the faster general correlator elsewhere
        h->wgroups.clear();
        int maxWP = 0;
        bool all_ok = true, any_ok = false;
        for (int b = 0; b < nbands; ) {
            int e = b + 1;
            while (e < nbands && h->W[e] == h->W[b]) ++e;
            nbls_wgroup g{b, e, h->W[b], h->unit_off[b], h->unit_off[e], false};
            int S_, PFB_, CSB_, CSA_, WP_, nsl_, G_, NC_;
            size_t lds_;
            g.screen = h->d_xij && nbls_screen_geometry(h, g.W, &S_, &PFB_, &CSB_, &CSA_, &WP_, &lds_, &nsl_, &G_, &NC_);
            if (g.screen) { any_ok = true; if (WP_ > maxWP) maxWP = WP_; }
            else if (g.u1 > g.u0) all_ok = false;
            h->wgroups.push_back(g);
            b = e;
        }
        if (xcorr_impl == 3 && !all_ok)
            return fail(h, NBLS_ERR_UNSUPPORTED, "nbls_plan: the int8 screening correlator needs 3..33 channels and channel images that fit a CU's LDS");
        if (xcorr_impl == 0 && any_ok) xcorr_impl = 3;
        h->xcorr_impl = xcorr_impl;
        if (xcorr_impl != 3) for (nbls_wgroup& g : h->wgroups) g.screen = false;
        h->screen_wp = maxWP;
    }
    if (xcorr_impl == 3) {
        const int WP_ = h->screen_wp;
        // unit batches small enough for the quantised windows to stay in the 256 MiB Infinity Cache (192 MB: measured
        // against 96 — the value of rounds 1-2 — and 384 at every BASELINE shape: 0.2-1.2 % of the pass, fewer launch tails)
        const int64_t batch_mb = h->opt.screen_batch_mb > 0 ? h->opt.screen_batch_mb : 192;
        int64_t batch = (int64_t)(batch_mb << 20) / ((int64_t)h->nchans * 2 * WP_);
        if (batch < 64) batch = 64;
        if (batch > U) batch = U > 0 ? U : 1;
        // equal batches (a multiple of 8 units, the XCD grouping of the screening grid) instead of full ones plus a
        // remainder: a 200-unit tail batch pays four kernel launches and their drain for next to nothing
        if (U > batch) {
            const int64_t nb_ = (U + batch - 1) / batch;
            int64_t eq = ((U + nb_ - 1) / nb_ + 7) / 8 * 8;
            if (eq < batch) batch = eq;
        }
        h->screen_batch = batch;
        if ((rc = ensure(h, &h->d_qbuf, &h->cap_qbuf, (size_t)batch * h->nchans * 2 * WP_))) return rc;
        if ((rc = ensure(h, &h->d_qmeta, &h->cap_qmeta, (size_t)batch * h->nchans * (10 + WP_ / 32) * sizeof(double)))) return rc;
        if (h->opt.screen_stamps || h->opt.lts_stamps) {
            if ((rc = ensure(h, &h->d_stamps, &h->cap_stamps, (size_t)(batch + 8) * h->nchans * ((h->nchans + 1) / 2) * 8 * sizeof(unsigned long long)))) return rc;   // one record per screening workgroup: (unit, sliding channel, partner group)
        }
        if ((rc = ensure(h, &h->d_cand, &h->cap_cand, (size_t)batch * h->nchans * h->nchans * 32 * sizeof(int32_t)))) return rc;
    }
    h->lts = lts != nullptr;
    if (lts) {
        h->ltsp = *lts;
        h->ltsp.starts = nullptr;
        h->ltsp.rew_table = nullptr;
        if ((rc = alloc_copy(h, &h->d_starts, lts->starts, (size_t)lts->nstarts * 4))) return rc;
        if ((rc = alloc_copy(h, &h->d_rew, lts->rew_table, (size_t)P + 1))) return rc;
        const int PP = P + 16;                           // padding: the large-array LTS kernel fetches one block of pairs ahead
        std::vector<double> xs((size_t)PP * 2, 0.0), xc((size_t)PP, 0.0);
        for (int k = 0; k < P; ++k) {
            xs[2 * k] = h->h_xij[2 * k] / lts->xij_mad[0];
            xs[2 * k + 1] = h->h_xij[2 * k + 1] / lts->xij_mad[1];
            xc[k] = xs[2 * k] * xs[2 * k + 1];
        }
        if ((rc = alloc_copy(h, &h->d_xs, xs.data(), xs.size()))) return rc;
        if ((rc = alloc_copy(h, &h->d_xc, xc.data(), xc.size()))) return rc;
        const int NS = (P + 3) / 4;                      // every 4th pair: the sample pass of the large-array LTS kernel
        std::vector<double> xss((size_t)(NS + 16) * 2, 0.0);
        for (int i = 0; i < NS; ++i) { xss[2 * i] = xs[2 * (4 * i)]; xss[2 * i + 1] = xs[2 * (4 * i) + 1]; }
        if ((rc = alloc_copy(h, &h->d_xss, xss.data(), xss.size()))) return rc;
    }
    // the arena's tables in one piece (copies queued before it on the same stream came from other parts of the staging arena)
    h->arena_mode = false;
    if (h->stage && h->d_parena && h->stage_used)
        HIPCHK(h, hipMemcpyAsync(h->d_parena, h->stage, h->stage_used, hipMemcpyHostToDevice, h->up));
    h->planned = true;
    if (h->opt.plan_timing) {
        const auto tp3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "nbls_plan: tables %.3f ms, uploads %.3f ms, buffers+rest %.3f ms\n", ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3));
    }
    return NBLS_OK;
}

int nbls_execute(nbls_handle* h) { return nbls_execute_stages(h, 7); }

int nbls_execute_after(nbls_handle* h, nbls_handle* prev) {
    if (!h || !prev || prev == h) return NBLS_ERR_ARG;
    if (prev->device != h->device) return fail(h, NBLS_ERR_ARG, "nbls_execute_after: the two handles are on different devices");
    if (!prev->ev_xd || !prev->ev_xd_recorded) return fail(h, NBLS_ERR_STATE, "nbls_execute_after: the other handle has not queued a pass");
    h->after = prev;
    const int rc = nbls_execute_stages(h, 7);
    h->after = nullptr;
    return rc;
}

int nbls_execute_stages(nbls_handle* h, int32_t stage_mask) {
    if (!h) return NBLS_ERR_ARG;
    if (!h->planned) return fail(h, NBLS_ERR_STATE, "nbls_execute: no plan");
    if ((stage_mask & 6) && !h->d_xij) return fail(h, NBLS_ERR_STATE, "nbls_execute: no geometry set");
    // the samples: uploaded — or on their way (nbls_upload_rows on another thread): a pass with a filter stage then takes
    // the channels as they land; any other pass needs them all before it is queued
    const int upload_st = h->upload_state.load(std::memory_order_acquire);      // (read BEFORE trace_loaded: the upload thread sets that first)
    if (!((stage_mask & 1) && (upload_st == 1 || upload_st == 3)) && !h->trace_loaded)
        return fail(h, NBLS_ERR_STATE, "nbls_execute: the trace was declared (nbls_set_trace_shape) but not uploaded");
    HIPCHK(h, hipSetDevice(h->device));
    wait_uploads(h);
    h->work_queued = true;
    const size_t cells = (size_t)h->nbands * h->vector_len;
    const int P = h->d_xij ? h->npairs : 1;
    // padding beyond nwin[b] is zeros (narrow_band_least_squares.py:268-272): the result block (grids + weight mask) is
    // cleared here; the per-pair side arrays (lag, cmax, weights, z: 25 MB at cfg-3, five fill kernels per band group) are
    // written for every computed window and read for no other — the rows nobody computed are zeroed on the host by the
    // rare caller that fetches them (nbls_fetch: zero_uncomputed)
    (void)cells; (void)P;
    // (a caller that did not wait for every streamed batch of the previous pass: its copies still read the block)
    if (!h->rbatches.empty() && h->cstream) HIPCHK(h, hipStreamWaitEvent(h->stream, h->rev[2 * (h->rbatches.size() - 1) + 1], 0));
    h->rbatches.clear();
    HIPCHK(h, hipMemsetAsync(h->d_res, 0, h->res_bytes, h->stream));
    if (h->prof) HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    if (stage_mask & 1) {
        // every (band, channel) series is filtered on its own: the channels that have landed so far, then the next ones
        // as their events are recorded (one launch of all channels when the trace is there already)
        for (int done = 0; done < h->nchans;) {
            int landed = wait_rows(h, done);
            if (landed < 0 || landed <= done)
                return fail(h, NBLS_ERR_STATE, landed == -2 ? "nbls_execute: the announced trace did not arrive (no nbls_upload_rows within 120 s)"
                                               : landed == -3 ? "nbls_execute: the trace was declared (nbls_set_trace_shape) but not uploaded"
                                                              : "nbls_execute: the upload of the trace failed");
            if (h->opt.filter_row_step > 0 && landed > done + h->opt.filter_row_step) landed = done + h->opt.filter_row_step;   // (A/B and tests)
            if ((int)h->uev.size() >= landed) HIPCHK(h, hipStreamWaitEvent(h->stream, h->uev[landed - 1], 0));
            HIPCHK(h, nbls_launch_filter(h, done, landed - done));
            done = landed;
        }
    }
    if (h->prof) HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    h->solve_done = false;
    h->last_stage_mask = stage_mask;
    // per-batch solves: behind each unit batch of the correlation stage (streamed results: a batch's rows are complete
    // while later batches are still being correlated), on the second stream with option "overlap"
    h->fuse_solve = ((stage_mask & 6) == 6) && h->xcorr_impl == 3 && (h->opt.overlap > 0 || h->stream_results);
    // option "overlap": 1 on, -1 off, 0 auto = on for a streamed pass of several SMALL unit batches (under 1024 units on
    // average: adaptive windows give every window length a batch of its own, and a batch of a few dozen units is all
    // launch latency and drain — device pass of cfg-1b 0.95 -> 0.75 ms).  Big batches gain 1.5-3 % (cfg-3 14.7 -> 14.3 ms,
    // cfg-5 174.6 -> 169.0) at the price of kernels whose measured durations include their neighbour's: left off there.
    int64_t nbatch_est = 0;
    for (const nbls_wgroup& g : h->wgroups)
        if (g.u1 > g.u0) nbatch_est += g.screen && h->screen_batch > 0 ? (g.u1 - g.u0 + h->screen_batch - 1) / h->screen_batch : 1;
    const bool small_batches = nbatch_est > 1 && h->nunits / nbatch_est < 1024;
    h->solve_on_stream2 = h->fuse_solve && h->stream2 && (h->opt.overlap > 0 || (h->opt.overlap == 0 && h->stream_results && small_batches));
    if (h->stream_results) {
        if (!h->cstream) HIPCHK(h, hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking));
        if (h->cap_hres < h->res_bytes) {
            HIPCHK(h, hipStreamSynchronize(h->cstream));
            if (h->h_res) { (void)hipHostFree(h->h_res); h->h_res = nullptr; h->cap_hres = 0; }
            HIPCHK(h, hipHostMalloc((void**)&h->h_res, h->res_bytes ? h->res_bytes : 8, hipHostMallocDefault));
            h->cap_hres = h->res_bytes ? h->res_bytes : 8;
        }
    }
    // nbls_execute_after: this pass's filter may run beside the other handle's correlation stage (memory-bound next to
    // matrix-core-bound), its own correlation stage starts when the other one's is through
    if (h->after && (stage_mask & 2)) HIPCHK(h, hipStreamWaitEvent(h->stream, h->after->ev_xd, 0));
    h->ev_xd_by_launcher = false;
    if (stage_mask & 2) HIPCHK(h, nbls_launch_xcorr(h));
    if (h->ev_xd && (stage_mask & 2) && !h->ev_xd_by_launcher) { HIPCHK(h, hipEventRecord(h->ev_xd, h->stream)); h->ev_xd_recorded = true; }
    if (h->prof) HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
    if ((stage_mask & 4) && !h->solve_done) HIPCHK(h, nbls_launch_solve(h));
    if (h->prof) { HIPCHK(h, hipEventRecord(h->ev[3], h->stream)); h->ev_valid = true; }
    // streamed results of a pass that was not cut into batches (general correlators, stage subsets, no units): ONE batch
    if (h->stream_results && h->rbatches.empty()) HIPCHK(h, nbls_queue_result_batch(h, 0, h->nunits, h->stream));
    return NBLS_OK;
}

}  // extern "C"

// The rows of units [u0, u1) are complete once `producer` has run what is queued on it: their part of the result block
// (the cell range [c0, c1) of each of the four grids and of the mask; padding cells between two bands ride along, they
// are zero) goes to the pinned mirror on the copy stream, and an event marks the landing (nbls_wait_result_batch).
hipError_t nbls_queue_result_batch(nbls_handle* h, int64_t u0, int64_t u1, hipStream_t producer) {
    if (!h->stream_results || !h->cstream || !h->h_res) return hipSuccess;
    const size_t k = h->rbatches.size();
    while (h->rev.size() < 2 * (k + 1)) {
        hipEvent_t e;
        hipError_t ce = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (ce != hipSuccess) return ce;
        h->rev.push_back(e);
    }
    nbls_handle::result_batch rb{u0, u1, 0, 0};
    if (u1 > u0) {
        auto cell = [&](int64_t u) {
            const int b = (int)(std::upper_bound(h->unit_off.begin(), h->unit_off.end(), (int32_t)u) - h->unit_off.begin()) - 1;
            const int64_t first = (int)h->woff.size() == h->nbands ? h->woff[b] : 0;
            return (int64_t)b * h->vector_len + first + (u - h->unit_off[b]);
        };
        rb.c0 = cell(u0);
        rb.c1 = cell(u1 - 1) + 1;
    }
    hipError_t e = hipEventRecord(h->rev[2 * k], producer);
    if (e != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(h->cstream, h->rev[2 * k], 0)) != hipSuccess) return e;
    if (rb.c1 > rb.c0) {
        const size_t cells = (size_t)h->nbands * h->vector_len;
        for (int g = 0; g < 4; ++g) {
            const size_t off = ((size_t)g * cells + (size_t)rb.c0) * sizeof(double);
            if ((e = hipMemcpyAsync(h->h_res + off, h->d_res + off, (size_t)(rb.c1 - rb.c0) * sizeof(double), hipMemcpyDeviceToHost, h->cstream)) != hipSuccess) return e;
        }
        const size_t moff = 4 * cells * sizeof(double) + (size_t)rb.c0 * h->mask_bytes;
        if ((e = hipMemcpyAsync(h->h_res + moff, h->d_res + moff, (size_t)(rb.c1 - rb.c0) * h->mask_bytes, hipMemcpyDeviceToHost, h->cstream)) != hipSuccess) return e;
    }
    if ((e = hipEventRecord(h->rev[2 * k + 1], h->cstream)) != hipSuccess) return e;
    h->rbatches.push_back(rb);
    return hipSuccess;
}

extern "C" {

int nbls_stream_results(nbls_handle* h, int32_t on) {
    if (!h) return NBLS_ERR_ARG;
    h->stream_results = on != 0;
    return NBLS_OK;
}

int nbls_result_batches(nbls_handle* h, int32_t* nbatches) {
    if (!h || !nbatches) return NBLS_ERR_ARG;
    *nbatches = h->stream_results ? (int32_t)h->rbatches.size() : 0;
    return NBLS_OK;
}

int nbls_wait_result_batch(nbls_handle* h, int32_t k, int64_t* out4, const void** host_block) {
    if (!h) return NBLS_ERR_ARG;
    if (!h->stream_results || k < 0 || (size_t)k >= h->rbatches.size())
        return fail(h, NBLS_ERR_STATE, "nbls_wait_result_batch: no such batch (nbls_stream_results + nbls_execute first)");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventSynchronize(h->rev[2 * (size_t)k + 1]));
    const nbls_handle::result_batch& rb = h->rbatches[(size_t)k];
    if (out4) { out4[0] = rb.u0; out4[1] = rb.u1; out4[2] = rb.c0; out4[3] = rb.c1; }
    if (host_block) *host_block = h->h_res;
    return NBLS_OK;
}

// Wait for the handle's stream and, when profiling is on, turn the events of the last pass into timings.
static int finish_pass(nbls_handle* h) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->cstream && !h->rbatches.empty()) HIPCHK(h, hipStreamSynchronize(h->cstream));
    h->work_queued = false;
    if (h->prof && h->ev_valid) {
        float f = 0, x = 0, s = 0, t = 0;
        HIPCHK(h, hipEventElapsedTime(&f, h->ev[0], h->ev[1]));
        HIPCHK(h, hipEventElapsedTime(&x, h->ev[1], h->ev[2]));
        HIPCHK(h, hipEventElapsedTime(&s, h->ev[2], h->ev[3]));
        HIPCHK(h, hipEventElapsedTime(&t, h->ev[0], h->ev[3]));
        h->tim.filter_ms = f; h->tim.xcorr_ms = x; h->tim.solve_ms = s; h->tim.total_ms = t;
        h->tim.quantize_ms = h->tim.screen_ms = h->tim.verify_ms = 0.0;
        double fused_solve = 0.0;
        for (int b = 0; b + 4 < h->bev_used; b += 5) {
            float q = 0, sc = 0, v = 0, so = 0;
            HIPCHK(h, hipEventElapsedTime(&q, h->bev[b], h->bev[b + 1]));
            HIPCHK(h, hipEventElapsedTime(&sc, h->bev[b + 1], h->bev[b + 2]));
            HIPCHK(h, hipEventElapsedTime(&v, h->bev[b + 2], h->bev[b + 3]));
            if (h->prof_fused) HIPCHK(h, hipEventElapsedTime(&so, h->bev[b + 3], h->bev[b + 4]));
            h->tim.quantize_ms += q; h->tim.screen_ms += sc; h->tim.verify_ms += v; fused_solve += so;
        }
        if (h->prof_fused && h->bev_used > 0) {
            // per-batch solves sit inside the correlation stage's interval: report the stages as if they were separate
            // (with option "overlap" the solves run beside the next batch's correlation: their sum is not wall time)
            h->tim.solve_ms += fused_solve;
            if (!h->solve_on_stream2) h->tim.xcorr_ms -= fused_solve;
        }
        h->bev_used = 0;
        h->tim.xcorr_impl = h->xcorr_impl_used;
        h->ev_valid = false;
    }
    return NBLS_OK;
}

int nbls_sync(nbls_handle* h) {
    if (!h) return NBLS_ERR_ARG;
    return finish_pass(h);
}

int nbls_fetch(nbls_handle* h, double* vel, double* baz, double* mdccm, double* sigma_tau, int32_t* nwin,
               int32_t* lag, double* cmax, uint8_t* weights, double* z) {
    if (!h) return NBLS_ERR_ARG;
    if (!h->planned) return fail(h, NBLS_ERR_STATE, "nbls_fetch: no plan");
    { const int rc = finish_pass(h); if (rc) return rc; }
    const size_t cells = (size_t)h->nbands * h->vector_len;
    const double* dg[4] = {h->d_vel, h->d_baz, h->d_mdccm, h->d_sig};
    double* hg[4] = {vel, baz, mdccm, sigma_tau};
    if (vel && baz == vel + cells && mdccm == baz + cells && sigma_tau == mdccm + cells) {
        HIPCHK(h, copy_sync(h, vel, h->d_vel, 4 * cells * sizeof(double), hipMemcpyDeviceToHost));   // caller's grids are one block too
    } else {
        for (int g = 0; g < 4; ++g)
            if (hg[g]) HIPCHK(h, copy_sync(h, hg[g], dg[g], cells * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (nwin) memcpy(nwin, h->nwin.data(), h->nbands * sizeof(int32_t));
    // rows of windows this plan did not compute (beyond a band's count, outside a window slice) are zeros
    auto zero_uncomputed = [&](void* out, size_t row_bytes) {
        unsigned char* o = (unsigned char*)out;
        for (int b = 0; b < h->nbands; ++b) {
            const int64_t first = (int)h->woff.size() == h->nbands ? h->woff[b] : 0, n = h->nwin[b];
            unsigned char* band = o + (size_t)b * h->vector_len * row_bytes;
            if (first > 0) memset(band, 0, (size_t)first * row_bytes);
            if (first + n < h->vector_len) memset(band + (size_t)(first + n) * row_bytes, 0, (size_t)(h->vector_len - first - n) * row_bytes);
        }
    };
    // the side arrays are not cleared by a pass (see nbls_execute_stages): what a stage that did NOT run in the last pass
    // would have written is zeros here, not the previous pass's values
    const bool ran_x = (h->last_stage_mask & 2) != 0, ran_s = (h->last_stage_mask & 4) != 0;
    if (lag) {
        if (ran_x) { HIPCHK(h, copy_sync(h, lag, h->d_lag, cells * h->npairs * sizeof(int32_t), hipMemcpyDeviceToHost)); zero_uncomputed(lag, h->npairs * sizeof(int32_t)); }
        else memset(lag, 0, cells * h->npairs * sizeof(int32_t));
    }
    if (cmax) {
        if (ran_x) { HIPCHK(h, copy_sync(h, cmax, h->d_cmax, cells * h->npairs * sizeof(double), hipMemcpyDeviceToHost)); zero_uncomputed(cmax, h->npairs * sizeof(double)); }
        else memset(cmax, 0, cells * h->npairs * sizeof(double));
    }
    if (weights) {
        if (ran_s) { HIPCHK(h, copy_sync(h, weights, h->d_wts, cells * h->npairs, hipMemcpyDeviceToHost)); zero_uncomputed(weights, (size_t)h->npairs); }
        else memset(weights, 0, cells * h->npairs);
    }
    if (z) {
        if (ran_s) { HIPCHK(h, copy_sync(h, z, h->d_z, 2 * cells * sizeof(double), hipMemcpyDeviceToHost)); zero_uncomputed(z, 2 * sizeof(double)); }
        else memset(z, 0, 2 * cells * sizeof(double));
    }
    return NBLS_OK;
}

int nbls_set_uncertainty(nbls_handle* h, const double* eig6) {
    if (!h) return NBLS_ERR_ARG;
    h->want_unc = eig6 != nullptr;
    if (eig6) {
        if (!(eig6[0] > 0.0) || !(eig6[1] > 0.0)) return fail(h, NBLS_ERR_ARG, "nbls_set_uncertainty: the eigenvalues of X^T X must be positive");
        for (int i = 0; i < 6; ++i) h->unc_par[i] = eig6[i];
    }
    h->planned = false;                      // (the next plan sizes the output buffer)
    return NBLS_OK;
}

int nbls_fetch_uncertainty(nbls_handle* h, double* vel_uncert, double* baz_uncert) {
    if (!h) return NBLS_ERR_ARG;
    if (!h->planned) return fail(h, NBLS_ERR_STATE, "nbls_fetch_uncertainty: no plan");
    if (!h->want_unc || !h->d_unc) return fail(h, NBLS_ERR_STATE, "nbls_fetch_uncertainty: nbls_set_uncertainty before nbls_plan");
    { const int rc = finish_pass(h); if (rc) return rc; }
    const size_t cells = (size_t)h->nbands * h->vector_len;
    double* outs[2] = {vel_uncert, baz_uncert};
    for (int g = 0; g < 2; ++g) {
        if (!outs[g]) continue;
        if (!(h->last_stage_mask & 4)) { memset(outs[g], 0, cells * sizeof(double)); continue; }
        HIPCHK(h, copy_sync(h, outs[g], h->d_unc + g * cells, cells * sizeof(double), hipMemcpyDeviceToHost));
        // rows of windows this plan did not compute are zeros, like the grids
        for (int b = 0; b < h->nbands; ++b) {
            const int64_t first = (int)h->woff.size() == h->nbands ? h->woff[b] : 0, n = h->nwin[b];
            double* band = outs[g] + (size_t)b * h->vector_len;
            for (int64_t w = 0; w < first; ++w) band[w] = 0.0;
            for (int64_t w = first + n; w < h->vector_len; ++w) band[w] = 0.0;
        }
    }
    return NBLS_OK;
}

int nbls_fetch_filtered(nbls_handle* h, int32_t band, double* out) {
    if (!h || !out) return NBLS_ERR_ARG;
    if (!h->planned) return fail(h, NBLS_ERR_STATE, "nbls_fetch_filtered: no plan");
    if (band < 0 || band >= h->nbands) return fail(h, NBLS_ERR_ARG, "nbls_fetch_filtered: band out of range");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy2DAsync(out, h->npts * sizeof(double),
                               h->d_filt + (size_t)band * h->nchans * h->npts_pad, h->npts_pad * sizeof(double),
                               h->npts * sizeof(double), h->nchans, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return NBLS_OK;
}

int nbls_set_filtered(nbls_handle* h, int32_t band, const double* data) {
    if (!h || !data) return NBLS_ERR_ARG;
    if (!h->planned) return fail(h, NBLS_ERR_STATE, "nbls_set_filtered: no plan");
    if (band < 0 || band >= h->nbands) return fail(h, NBLS_ERR_ARG, "nbls_set_filtered: band out of range");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy2DAsync(h->d_filt + (size_t)band * h->nchans * h->npts_pad, h->npts_pad * sizeof(double), data,
                               h->npts * sizeof(double), h->npts * sizeof(double), h->nchans, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return NBLS_OK;
}

int nbls_filter_segment(nbls_handle* h, int32_t reverse, const double* state_in, double* state_out) {
    if (!h) return NBLS_ERR_ARG;
    if (!h->planned) return fail(h, NBLS_ERR_STATE, "nbls_filter_segment: no plan");
    if (h->nsections < 1) return fail(h, NBLS_ERR_ARG, "nbls_filter_segment: the plan has no filter sections");
    if (state_out && (h->npts % NBLS_FILTER_CHUNK) != 0 && !reverse)
        return fail(h, NBLS_ERR_ARG, "nbls_filter_segment: a segment whose end state is wanted must be a whole number of 512-sample chunks");
    if (reverse && state_in && (h->npts % NBLS_FILTER_CHUNK) != 0)
        return fail(h, NBLS_ERR_ARG, "nbls_filter_segment: a backward segment that continues a state must be a whole number of 512-sample chunks");
    HIPCHK(h, hipSetDevice(h->device));
    wait_uploads(h);
    h->work_queued = true;
    const size_t n = (size_t)h->nbands * h->nchans * 2 * h->nsections;
    int rc;
    if ((rc = ensure(h, &h->d_seg_state, &h->cap_seg_state, 2 * n * sizeof(double)))) return rc;
    double* d_init = nullptr;
    double* d_fin = state_out ? h->d_seg_state + n : nullptr;
    if (state_in) {
        d_init = h->d_seg_state;
        HIPCHK(h, hipMemcpyAsync(d_init, state_in, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, nbls_launch_filter_segment(h, reverse ? 1 : 0, d_init, d_fin));
    if (state_out) HIPCHK(h, hipMemcpyAsync(state_out, d_fin, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return NBLS_OK;
}

int nbls_device_results(nbls_handle* h, void** ptrs, int64_t* bytes_per_grid) {
    if (!h || !ptrs) return NBLS_ERR_ARG;
    if (!h->planned) return fail(h, NBLS_ERR_STATE, "nbls_device_results: no plan");
    ptrs[0] = h->d_vel; ptrs[1] = h->d_baz; ptrs[2] = h->d_mdccm; ptrs[3] = h->d_sig; ptrs[4] = h->d_nwin;
    if (bytes_per_grid) *bytes_per_grid = (int64_t)h->nbands * h->vector_len * (int64_t)sizeof(double);
    return NBLS_OK;
}

int nbls_result_layout(nbls_handle* h, int64_t* out4) {
    if (!h || !out4) return NBLS_ERR_ARG;
    if (!h->planned) return fail(h, NBLS_ERR_STATE, "nbls_result_layout: no plan");
    out4[0] = (int64_t)h->nbands * h->vector_len;
    out4[1] = h->mask_bytes;
    out4[2] = (int64_t)h->res_bytes;
    out4[3] = (int64_t)(4 * sizeof(double)) * out4[0];      // byte offset of the mask
    return NBLS_OK;
}

int nbls_fetch_packed(nbls_handle* h, void* out, int64_t nbytes) {
    if (!h || !out) return NBLS_ERR_ARG;
    if (!h->planned) return fail(h, NBLS_ERR_STATE, "nbls_fetch_packed: no plan");
    if (nbytes != (int64_t)h->res_bytes) return fail(h, NBLS_ERR_ARG, "nbls_fetch_packed: size does not match nbls_result_layout");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(out, h->d_res, h->res_bytes, hipMemcpyDeviceToHost, h->stream));   // ordered after the pass
    return finish_pass(h);
}

int nbls_load_result_block(nbls_handle* h, const void* block, int64_t nbytes) {
    // the block a rank assembled on the host from several HBM rounds (nbls_fetch_packed per round) goes back into the
    // handle's result block, where nbls_comm_gather sends from
    if (!h || !block || nbytes < 0) return NBLS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t need = std::max((size_t)nbytes + 8, h->reserve_res);
    int rc;
    if ((rc = ensure(h, &h->d_res, &h->cap_res, need))) return rc;
    HIPCHK(h, hipMemsetAsync(h->d_res, 0, h->cap_res, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_res, block, (size_t)nbytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));              // the host buffer may go away
    h->res_bytes = (size_t)nbytes;
    h->d_vel = h->d_baz = h->d_mdccm = h->d_sig = nullptr;   // the views of the last plan no longer describe the block
    h->d_mask = nullptr;
    h->planned = false;                                      // ... and a new pass needs a new plan
    h->res_loaded = true;                                    // (nbls_comm_gather: this block is a result, whatever its allocation's size)
    return NBLS_OK;
}

int nbls_set_option(nbls_handle* h, const char* key, int64_t value) {
    if (!h || !key) return NBLS_ERR_ARG;
    struct Key { const char* name; int nbls_options::*field; bool developer; };
    static const Key keys[] = {
        {"lts_impl", &nbls_options::lts_impl, false},
        {"lts_generic_h", &nbls_options::lts_generic_h, false},
        {"lts_coop_threads", &nbls_options::lts_coop_threads, false},
        {"lts_sample_its", &nbls_options::lts_sample_its, false},
        {"screen_nsl1", &nbls_options::screen_nsl1, false},
        {"screen_static", &nbls_options::screen_static, false},
        {"screen_tb4", &nbls_options::screen_tb4, false},
        {"screen_pretest", &nbls_options::screen_pretest, false},
        {"screen_tb8", &nbls_options::screen_tb8, false},
        {"screen_cxx", &nbls_options::screen_cxx, false},
        {"screen_nc4", &nbls_options::screen_nc4, false},
        {"screen_batch_mb", &nbls_options::screen_batch_mb, false},
        {"overlap", &nbls_options::overlap, false},
        {"solve_min_units", &nbls_options::solve_min_units, false},
        {"result_tail_units", &nbls_options::result_tail_units, false},
        {"filter_row_step", &nbls_options::filter_row_step, false},
        {"filter_nofuse", &nbls_options::filter_nofuse, false},
        {"filter_nomfma", &nbls_options::filter_nomfma, false},
        {"ablate", &nbls_options::ablate, true},
        {"screen_stamps", &nbls_options::screen_stamps, true},
        {"screen_seed", &nbls_options::screen_seed, true},
        {"lts_stamps", &nbls_options::lts_stamps, true},
        {"screen_pad_kb", &nbls_options::screen_pad_kb, true},
        {"lts_pad_kb", &nbls_options::lts_pad_kb, true},
        {"plan_timing", &nbls_options::plan_timing, true},
    };
    if (strcmp(key, "stream_priority") == 0) {
        // 0 normal, > 0 lower, < 0 higher (clamped to what the device offers).  Several handles of one GPU running
        // passes side by side (the band groups of a pipelined call): the pass whose results the host wants first
        // gets its workgroups dispatched first.  The handle must be idle; results do not depend on it.
        HIPCHK(h, hipSetDevice(h->device));
        int least = 0, greatest = 0;
        HIPCHK(h, hipDeviceGetStreamPriorityRange(&least, &greatest));       // numerically: greatest <= least
        int prio = (int)value;
        if (prio < greatest) prio = greatest;
        if (prio > least) prio = least;
        if (prio == h->stream_priority) return NBLS_OK;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->stream2) HIPCHK(h, hipStreamSynchronize(h->stream2));
        hipStream_t s1 = nullptr, s2 = nullptr;
        HIPCHK(h, hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, prio));
        if (hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, prio) != hipSuccess) s2 = nullptr;
        if (h->stream2) (void)hipStreamDestroy(h->stream2);
        if (h->up == h->stream) h->up = s1;
        (void)hipStreamDestroy(h->stream);
        h->stream = s1;
        h->stream2 = s2;
        h->stream_priority = prio;
        return NBLS_OK;
    }
    for (const Key& k : keys) {
        if (strcmp(k.name, key) != 0) continue;
#ifndef NBLS_DEVELOPER
        if (k.developer) return fail(h, NBLS_ERR_UNSUPPORTED, std::string("option '") + key + "' exists only in the developer build (make dev)");
#endif
        h->opt.*(k.field) = (int)value;
        h->planned = false;                 // options are read by nbls_plan and by the launchers of the next pass
        return NBLS_OK;
    }
    return fail(h, NBLS_ERR_ARG, std::string("unknown option '") + key + "'");
}

int nbls_developer_build(void) {
#ifdef NBLS_DEVELOPER
    return 1;
#else
    return 0;
#endif
}

int nbls_set_profiling(nbls_handle* h, int32_t on) {
    if (!h) return NBLS_ERR_ARG;
    h->prof = on != 0;
    return NBLS_OK;
}

int nbls_get_timings(nbls_handle* h, nbls_timings* out) {
    if (!h || !out) return NBLS_ERR_ARG;
    *out = h->tim;
    return NBLS_OK;
}

int nbls_run(nbls_handle* h, int32_t nbands, const double* sos, int32_t nsections, int32_t zero_phase,
             const double* taper_left, const double* taper_right, int32_t taper_len, const int32_t* winlen,
             const int32_t* wininc, int32_t vector_len, const nbls_lts_params* lts, int32_t xcorr_impl,
             double* vel, double* baz, double* mdccm, double* sigma_tau, int32_t* nwin, int32_t* lag,
             double* cmax, uint8_t* weights, double* z) {
    int rc = nbls_plan(h, nbands, sos, nsections, zero_phase, taper_left, taper_right, taper_len, winlen,
                       wininc, vector_len, lts, xcorr_impl);
    if (rc) return rc;
    if ((rc = nbls_execute(h))) return rc;
    if ((rc = nbls_sync(h))) return rc;
    return nbls_fetch(h, vel, baz, mdccm, sigma_tau, nwin, lag, cmax, weights, z);
}

int nbls_probe_mfma_f64(nbls_handle* h, const double* a, const double* b, double* out) {
    if (!h || !a || !b || !out) return NBLS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    HIPCHK(h, hipMalloc((void**)&da, 64 * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&db, 64 * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&dout, 256 * sizeof(double)));
    HIPCHK(h, copy_sync(h, da, a, 64 * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, copy_sync(h, db, b, 64 * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, nbls_launch_probe_mfma(h, da, db, dout));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, copy_sync(h, out, dout, 256 * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return NBLS_OK;
}

int nbls_debug_screen_stats(nbls_handle* h, int64_t* out4) {
    // out4 = {ordered pairs in the last batch, pairs whose candidate buffer overflowed,
    //         total candidates, max candidates}  (developer statistic of the int8 screening path)
    if (!h || !out4) return NBLS_ERR_ARG;
    if (!h->d_cand || h->screen_batch <= 0) return fail(h, NBLS_ERR_STATE, "no screening run yet");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int N = h->nchans;
    const int64_t last = h->last_batch > 0 ? h->last_batch : h->nunits - ((h->nunits - 1) / h->screen_batch) * h->screen_batch;
    std::vector<int32_t> c((size_t)last * N * N * 32);
    HIPCHK(h, copy_sync(h, c.data(), h->d_cand, c.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int q = 0; q < 8; ++q) out4[q] = 0;
    for (int64_t u = 0; u < last; ++u)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                if (i == j) continue;
                const int32_t* e = &c[((u * N + i) * N + j) * 32];
                out4[0] += 1;
                out4[1] += e[1] != 0;          // interval and/or list overflow
                out4[2] += e[0];
                if (e[0] > out4[3]) out4[3] = e[0];
                // how the listed lags cluster (developer: what a verifier that shares loads between adjacent lags could use)
                int n = e[0] < 26 ? e[0] : 26;
                int v[26];
                for (int q = 0; q < n; ++q) v[q] = e[6 + q];
                std::sort(v, v + n);
                for (int q = 0; q < n; ) {
                    int r = q + 1;
                    while (r < n && v[r] == v[r - 1] + 1) ++r;
                    out4[4] += 1;                       // runs of consecutive lags
                    if (r - q >= 2) out4[5] += r - q;   // listed lags that sit in a run of two or more
                    q = r;
                }
            }
    return NBLS_OK;
}

int nbls_debug_screen_stamps(nbls_handle* h, double* out6) {
    // mean cycle counts of the screen kernel's phases over the workgroups of the last batch:
    // out[10] = {stage issue, stage barrier wait, compute (wave 0), wait for the other waves, merge+write, total,
    //            wave 0: cycles inside the K loops, in the epilogues, of those: conversion + maximum, shared-maximum round trip}
    if (!h || !out6) return NBLS_ERR_ARG;
    if (!h->d_stamps) return fail(h, NBLS_ERR_STATE, "run with NBLS_SCREEN_STAMPS=1");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int64_t last = h->last_batch > 0 ? h->last_batch : h->nunits - ((h->nunits - 1) / h->screen_batch) * h->screen_batch;
    const int64_t nwg = ((last + 7) / 8) * 8 * h->nchans;   // upper bound (one or two channels per workgroup)
    std::vector<unsigned long long> st((size_t)nwg * 8);
    HIPCHK(h, copy_sync(h, st.data(), h->d_stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < 10; ++i) out6[i] = 0.0;
    int64_t cnt = 0;
    for (int64_t g = 0; g < nwg; ++g) {
        const unsigned long long* s = &st[(size_t)g * 8];
        if (s[5] <= s[0] || s[5] - s[0] > 100000000ull) continue;
        for (int i = 0; i < 5; ++i) out6[i] += (double)(s[i + 1] - s[i]);
        out6[5] += (double)(s[5] - s[0]);
        out6[6] += (double)(s[6] & 0xffffffffull);    // wave 0: cycles inside the K loops | in the epilogues
        out6[7] += (double)(s[6] >> 32);
        out6[8] += (double)(s[7] & 0xffffffffull);    // of the epilogues: conversion + maximum | shared-maximum round trip
        out6[9] += (double)(s[7] >> 32);
        ++cnt;
    }
    for (int i = 0; i < 10; ++i) out6[i] /= (double)(cnt > 0 ? cnt : 1);
    return NBLS_OK;
}

int nbls_debug_lts_stamps(nbls_handle* h, double* out8) {
    if (!h || !out8) return NBLS_ERR_ARG;
    if (!h->d_stamps || h->lts_stamp_waves <= 0) return fail(h, NBLS_ERR_STATE, "run with NBLS_SCREEN_STAMPS=1 NBLS_LTS_STAMPS=1");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> st((size_t)h->lts_stamp_waves * 8);
    HIPCHK(h, copy_sync(h, st.data(), h->d_stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) out8[i] = 0.0;
    int64_t cnt = 0;
    for (int64_t g = 0; g < h->lts_stamp_waves; ++g) {
        const unsigned long long* s = &st[(size_t)g * 8];
        if (s[6] <= s[0] || s[6] - s[0] > 1000000000ull) continue;
        for (int i = 0; i < 6; ++i) out8[i] += (double)(s[i + 1] - s[i]);
        out8[7] += (double)(s[6] - s[0]);
        out8[6] += (double)s[7];              // finished entries entering the candidate peel
        ++cnt;
    }
    for (int i = 0; i < 8; ++i) out8[i] /= (double)(cnt > 0 ? cnt : 1);
    return NBLS_OK;
}

int nbls_debug_lts_coop_breakdown(nbls_handle* h, double* out8) {
    // developer: mean over the units of the large-array LTS kernel's C-step phases, thread 0's cycles in
    // {groups (selection + sums), subset merging, compaction}, the live entries summed over the iterations, and
    // wave 0's {selection passes, selection cycles, sums cycles, groups}
    if (!h || !out8) return NBLS_ERR_ARG;
    if (!h->d_stamps || h->lts_stamp_waves <= 0) return fail(h, NBLS_ERR_STATE, "developer build with option lts_stamps needed");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> st((size_t)h->lts_stamp_waves * 8);
    HIPCHK(h, copy_sync(h, st.data(), h->d_stamps + (size_t)h->lts_stamp_waves * 8, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) out8[i] = 0.0;
    for (int64_t g = 0; g < h->lts_stamp_waves; ++g)
        for (int i = 0; i < 8; ++i) out8[i] += (double)st[(size_t)g * 8 + i];
    for (int i = 0; i < 8; ++i) out8[i] /= (double)h->lts_stamp_waves;
    return NBLS_OK;
}

int nbls_probe_mfma_i8(nbls_handle* h, const int32_t* a, const int32_t* b, int32_t* out) {
    if (!h || !a || !b || !out) return NBLS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    int *da = nullptr, *db = nullptr, *dout = nullptr;
    HIPCHK(h, hipMalloc((void**)&da, 256 * sizeof(int)));
    HIPCHK(h, hipMalloc((void**)&db, 256 * sizeof(int)));
    HIPCHK(h, hipMalloc((void**)&dout, 256 * sizeof(int)));
    HIPCHK(h, copy_sync(h, da, a, 256 * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, copy_sync(h, db, b, 256 * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, nbls_launch_probe_mfma_i8(h, da, db, dout));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, copy_sync(h, out, dout, 256 * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return NBLS_OK;
}

}  // extern "C"
