// Multi-GPU result gather of libnbls_hip.so on RCCL (xGMI inside a node).
//
// Reference: the band-parallel variant narrow_band_least_squares_parallel()
// (narrow_band_least_squares.py:223-323) fans the bands out with joblib (:285) and collects every
// worker's result rows in the parent (:291-320).  Here bands (or window slices) are sharded over GPUs,
// each GPU leaves its result block (nbls_result_layout) in its own HBM, and ONE grouped RCCL
// operation moves the blocks: a gather to a root rank (grouped ncclSend/ncclRecv) or an all-gather.
// No data-path collective exists besides this one — bands never exchange values.
//
// Two ways to form the communicator, same gather call afterwards:
//   nbls_comm_init_all   one process drives all GPUs (one handle per device), ncclCommInitAll
//   nbls_comm_init_rank  one process per GPU (torch.distributed.run / mpirun / anything that sets a
//                        rank): rank 0 makes the id with nbls_comm_unique_id and hands the 128 bytes
//                        to the others by any side channel (the Python host uses a TCP socket)
//
// RCCL is resolved with dlopen at the first comm call, so the library itself has no link-time
// dependency on it and single-GPU users never load it.
#include "nbls_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>     // types and prototypes only: every call goes through the table below

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

std::mutex g_mu;
RcclApi g_api;
std::string g_lib_path;          // nbls_comm_set_library: a library to resolve the ten entry points from instead of librccl
bool g_allow_shared = false;     // ... and whether nbls_comm_init_all may take several handles of one device

int cfail(nbls_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

// returns nullptr and fills *why when RCCL cannot be loaded
RcclApi* rccl(std::string* why) {
    std::lock_guard<std::mutex> l(g_mu);
    if (g_api.lib) return &g_api;
    // nbls_comm_set_library: ONLY that library (the tests' loopback stand-in runs several ranks of one process on
    // one GPU: tests/c_caller/loopback_rccl.cpp); otherwise RCCL by its usual names.  No environment variable is read.
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    if (!g_lib_path.empty()) lib = dlopen(g_lib_path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    else
        for (const char* n : names)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) {
        const char* de = dlerror();
        if (why) *why = std::string("RCCL not found (dlopen ") + (g_lib_path.empty() ? "librccl.so.1" : g_lib_path.c_str()) + "): " + (de ? de : "");
        return nullptr;
    }
#define NBLS_SYM(field, name)                                              \
    g_api.field = (decltype(g_api.field))dlsym(lib, name);                 \
    if (!g_api.field) { if (why) *why = std::string("RCCL symbol missing: ") + name; dlclose(lib); return nullptr; }
    NBLS_SYM(GetUniqueId, "ncclGetUniqueId")
    NBLS_SYM(CommInitRank, "ncclCommInitRank")
    NBLS_SYM(CommInitAll, "ncclCommInitAll")
    NBLS_SYM(CommDestroy, "ncclCommDestroy")
    NBLS_SYM(AllGather, "ncclAllGather")
    NBLS_SYM(Send, "ncclSend")
    NBLS_SYM(Recv, "ncclRecv")
    NBLS_SYM(GroupStart, "ncclGroupStart")
    NBLS_SYM(GroupEnd, "ncclGroupEnd")
    NBLS_SYM(GetErrorString, "ncclGetErrorString")
#undef NBLS_SYM
    g_api.lib = lib;
    return &g_api;
}

#define HIPC(h, call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return cfail(h, e_ == hipErrorOutOfMemory ? NBLS_ERR_NOMEM : NBLS_ERR_HIP,             \
                         std::string(#call) + ": " + hipGetErrorString(e_));                       \
    } while (0)

#define NCCLC(h, api, call)                                                                        \
    do {                                                                                           \
        ncclResult_t r_ = (call);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return cfail(h, NBLS_ERR_COMM, std::string(#call) + ": " + (api)->GetErrorString(r_)); \
    } while (0)

}  // namespace

extern "C" {

int nbls_comm_set_library(const char* path, int32_t allow_shared_device) {
    std::lock_guard<std::mutex> l(g_mu);
    const std::string want = path ? path : "";
    if (g_api.lib && want != g_lib_path) return NBLS_ERR_STATE;      // already resolved from somewhere else
    g_lib_path = want;
    g_allow_shared = allow_shared_device != 0;
    return NBLS_OK;
}

int nbls_comm_unique_id(void* id, int32_t nbytes) {
    if (!id || nbytes < (int32_t)sizeof(ncclUniqueId)) return NBLS_ERR_ARG;
    std::string why;
    RcclApi* api = rccl(&why);
    if (!api) return NBLS_ERR_COMM;
    ncclUniqueId uid;
    if (api->GetUniqueId(&uid) != ncclSuccess) return NBLS_ERR_COMM;
    memcpy(id, &uid, sizeof(uid));
    return NBLS_OK;
}

int nbls_comm_destroy(nbls_handle* h) {
    if (!h) return NBLS_ERR_ARG;
    if (h->comm) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        RcclApi* api = rccl(nullptr);
        if (api) (void)api->CommDestroy((ncclComm_t)h->comm);
        h->comm = nullptr;
    }
    if (h->d_gather) { (void)hipFree(h->d_gather); h->d_gather = nullptr; h->cap_gather = 0; }
    h->comm_world = 1;
    h->comm_rank = 0;
    return NBLS_OK;
}

int nbls_comm_init_rank(nbls_handle* h, const void* id, int32_t world, int32_t rank) {
    if (!h || !id || world < 1 || rank < 0 || rank >= world) return cfail(h, NBLS_ERR_ARG, "nbls_comm_init_rank: bad argument");
    std::string why;
    RcclApi* api = rccl(&why);
    if (!api) return cfail(h, NBLS_ERR_COMM, why);
    (void)nbls_comm_destroy(h);
    HIPC(h, hipSetDevice(h->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    NCCLC(h, api, api->CommInitRank(&c, world, uid, rank));
    h->comm = c;
    h->comm_world = world;
    h->comm_rank = rank;
    return NBLS_OK;
}

int nbls_comm_init_all(nbls_handle* const* hs, int32_t n) {
    if (!hs || n < 1 || n > 64) return NBLS_ERR_ARG;
    for (int i = 0; i < n; ++i) if (!hs[i]) return NBLS_ERR_ARG;
    std::string why;
    RcclApi* api = rccl(&why);
    if (!api) return cfail(hs[0], NBLS_ERR_COMM, why);
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        (void)nbls_comm_destroy(hs[i]);
        devs[i] = hs[i]->device;
        // (RCCL itself refuses a device twice; nbls_comm_set_library(path, 1) is for the loopback stand-in of the tests)
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i] && !g_allow_shared)
                return cfail(hs[0], NBLS_ERR_ARG, "nbls_comm_init_all: two handles on the same device");
    }
    std::vector<ncclComm_t> comms(n, nullptr);
    NCCLC(hs[0], api, api->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        hs[i]->comm = comms[i];
        hs[i]->comm_world = n;
        hs[i]->comm_rank = i;
    }
    return NBLS_OK;
}

int nbls_reserve_results(nbls_handle* h, int64_t bytes) {
    if (!h || bytes < 0) return NBLS_ERR_ARG;
    h->reserve_res = (size_t)bytes;
    return NBLS_OK;
}

// "Nobody hangs": a rank whose local preparation fails (a result block that does not fit block_bytes, a plan made
// without nbls_reserve_results, a failed allocation of the receive side, a failed HIP call) still ENTERS the
// collective — with a zeroed stand-in block whose status word is non-zero — and reports its own error afterwards;
// its peers see the status word instead of blocking in the all-gather.  An RCCL call that fails inside the group
// is remembered, the group is closed all the same, and the error is returned then.  Only calls that cannot take
// part at all return early: bad arguments that every rank shares (block_bytes, root), or no communicator.
int nbls_comm_gather(nbls_handle* const* hs, int32_t n, int32_t root, int64_t block_bytes, int64_t status,
                     void* host_out, int64_t host_bytes) {
    if (!hs || n < 1) return NBLS_ERR_ARG;
    for (int i = 0; i < n; ++i) if (!hs[i]) return NBLS_ERR_ARG;
    nbls_handle* h0 = hs[0];
    std::string why;
    RcclApi* api = rccl(&why);
    if (!api) return cfail(h0, NBLS_ERR_COMM, why);
    const int world = h0->comm_world;
    if (block_bytes < 16 || (block_bytes & 7)) return cfail(h0, NBLS_ERR_ARG, "nbls_comm_gather: block_bytes must be a multiple of 8");
    if (root >= world) return cfail(h0, NBLS_ERR_ARG, "nbls_comm_gather: root out of range");
    const size_t total = (size_t)world * (size_t)block_bytes;
    // which local handle delivers to the host: the root (gather) or the first one (all-gather)
    nbls_handle* deliver = nullptr;
    for (int i = 0; i < n; ++i) {
        nbls_handle* h = hs[i];
        if (!h->comm || h->comm_world != world) return cfail(h0, NBLS_ERR_STATE, "nbls_comm_gather: no communicator (nbls_comm_init_*)");
        if (root < 0 ? i == 0 : h->comm_rank == root) deliver = h;
    }
    int first_rc = NBLS_OK;                       // the first local failure: returned AFTER the collective
    auto note = [&](nbls_handle* h, int code, const std::string& msg) {
        if (first_rc == NBLS_OK) { first_rc = code; (void)cfail(h, code, msg); if (h != h0) (void)cfail(h0, code, msg); }
    };
    auto hipnote = [&](nbls_handle* h, hipError_t e, const char* what) {
        if (e != hipSuccess) note(h, e == hipErrorOutOfMemory ? NBLS_ERR_NOMEM : NBLS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
        return e == hipSuccess;
    };
    if (deliver && host_out && host_bytes != (int64_t)total)
        note(h0, NBLS_ERR_ARG, "nbls_comm_gather: host buffer must hold world * block_bytes");
    // ---- local preparation: every handle ends up with a send pointer of block_bytes, whatever went wrong ----
    std::vector<unsigned char*> send(n, nullptr);
    std::vector<unsigned char*> standin(n, nullptr);          // freed at the end
    for (int i = 0; i < n; ++i) {
        nbls_handle* h = hs[i];
        bool ok = hipnote(h, hipSetDevice(h->device), "hipSetDevice");
        bool use_own = ok;
        if (ok && (h->planned || h->res_loaded) && (int64_t)h->res_bytes + 8 > block_bytes) {
            note(h, NBLS_ERR_ARG, "nbls_comm_gather: block_bytes smaller than the result block + status word");
            use_own = false;
        }
        if (ok && use_own && h->res_loaded && h->d_res && h->cap_res < (size_t)block_bytes) {
            // a block assembled on the host (nbls_load_result_block) without nbls_reserve_results(block_bytes): it is
            // this rank's RESULT, not a failed rank's leftovers — move it into an allocation of the gather's size
            unsigned char* nb_ = nullptr;
            if (hipnote(h, hipMalloc((void**)&nb_, (size_t)block_bytes), "hipMalloc(result block)")) {
                (void)hipnote(h, hipMemsetAsync(nb_, 0, (size_t)block_bytes, h->stream), "hipMemsetAsync");
                (void)hipnote(h, hipMemcpyAsync(nb_, h->d_res, h->res_bytes, hipMemcpyDeviceToDevice, h->stream), "hipMemcpyAsync(loaded block)");
                (void)hipnote(h, hipStreamSynchronize(h->stream), "hipStreamSynchronize");
                (void)hipFree(h->d_res);
                h->d_res = nb_;
                h->cap_res = (size_t)block_bytes;
            } else use_own = false;
        }
        if (ok && use_own && (!h->d_res || h->cap_res < (size_t)block_bytes)) {
            if (h->planned) {
                note(h, NBLS_ERR_STATE, "nbls_comm_gather: call nbls_reserve_results(block_bytes) before nbls_plan");
                use_own = false;
            } else {
                // a rank that failed before it could plan: an empty block of its own
                if (h->d_res) { (void)hipFree(h->d_res); h->d_res = nullptr; h->cap_res = 0; }
                if (hipnote(h, hipMalloc((void**)&h->d_res, (size_t)block_bytes), "hipMalloc(result block)")) {
                    h->cap_res = (size_t)block_bytes;
                    (void)hipnote(h, hipMemsetAsync(h->d_res, 0, (size_t)block_bytes, h->stream), "hipMemsetAsync");
                } else {
                    h->d_res = nullptr;
                    use_own = false;
                }
            }
        }
        if (use_own) {
            send[i] = h->d_res;
        } else if (hipMalloc((void**)&standin[i], (size_t)block_bytes) == hipSuccess) {
            (void)hipMemsetAsync(standin[i], 0, (size_t)block_bytes, h->stream);
            send[i] = standin[i];
        } else {
            (void)hipGetLastError();
            // nothing to send from: this rank cannot take part (its peers will wait for it)
            for (int j = 0; j <= i; ++j) if (standin[j]) (void)hipFree(standin[j]);
            return cfail(h, NBLS_ERR_NOMEM, "nbls_comm_gather: no memory for a " + std::to_string(block_bytes) + "-byte block");
        }
        const bool recv_side = root < 0 || h->comm_rank == root;
        if (recv_side && (!h->d_gather || h->cap_gather < total)) {
            if (h->d_gather) { (void)hipFree(h->d_gather); h->d_gather = nullptr; h->cap_gather = 0; }
            if (hipnote(h, hipMalloc((void**)&h->d_gather, total), "hipMalloc(gather buffer)")) h->cap_gather = total;
            else {
                h->d_gather = nullptr;
                for (int j = 0; j <= i; ++j) if (standin[j]) (void)hipFree(standin[j]);
                return first_rc;                   // no receive buffer: cannot take part
            }
        }
    }
    // the status word: the caller's, or this process's first failure (every local block carries it)
    for (int i = 0; i < n; ++i) {
        nbls_handle* h = hs[i];
        (void)hipSetDevice(h->device);
        h->gather_status = first_rc != NBLS_OK ? (int64_t)first_rc : status;
        (void)hipnote(h, hipMemcpyAsync(send[i] + block_bytes - 8, &h->gather_status, 8, hipMemcpyHostToDevice, h->stream), "hipMemcpyAsync(status word)");
    }
    // ---- ONE grouped operation over all local ranks; the group is closed whatever happens inside ----
    auto ncclnote = [&](nbls_handle* h, ncclResult_t r, const char* what) {
        if (r != ncclSuccess) note(h, NBLS_ERR_COMM, std::string(what) + ": " + api->GetErrorString(r));
    };
    ncclnote(h0, api->GroupStart(), "ncclGroupStart");
    for (int i = 0; i < n; ++i) {
        nbls_handle* h = hs[i];
        ncclComm_t c = (ncclComm_t)h->comm;
        if (root < 0) {
            ncclnote(h, api->AllGather(send[i], h->d_gather, (size_t)block_bytes, ncclUint8, c, h->stream), "ncclAllGather");
        } else if (h->comm_rank == root) {
            for (int peer = 0; peer < world; ++peer) {
                if (peer == root) continue;
                ncclnote(h, api->Recv(h->d_gather + (size_t)peer * block_bytes, (size_t)block_bytes, ncclUint8, peer, c, h->stream), "ncclRecv");
            }
        } else {
            ncclnote(h, api->Send(send[i], (size_t)block_bytes, ncclUint8, root, c, h->stream), "ncclSend");
        }
    }
    ncclnote(h0, api->GroupEnd(), "ncclGroupEnd");
    for (int i = 0; i < n; ++i) {
        nbls_handle* h = hs[i];
        (void)hipSetDevice(h->device);
        if (root >= 0 && h->comm_rank == root)      // the root's own block: device-to-device, same stream
            (void)hipnote(h, hipMemcpyAsync(h->d_gather + (size_t)root * block_bytes, send[i], (size_t)block_bytes, hipMemcpyDeviceToDevice, h->stream), "hipMemcpyAsync(root block)");
        if (h == deliver && host_out && host_bytes == (int64_t)total)
            (void)hipnote(h, hipMemcpyAsync(host_out, h->d_gather, total, hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync(gathered blocks)");
    }
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(hs[i]->device);
        (void)hipnote(hs[i], hipStreamSynchronize(hs[i]->stream), "hipStreamSynchronize");
        if (standin[i]) (void)hipFree(standin[i]);
    }
    return first_rc;
}

}  // extern "C"
