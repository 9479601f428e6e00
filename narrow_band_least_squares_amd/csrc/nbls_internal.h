// Internal declarations shared by the translation units of libnbls_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "../../include/nbls.h"

#define NBLS_MAX_SECTIONS 8
#define NBLS_FILTER_CHUNK 512      // samples per scan chunk (one lane each)
#define NBLS_FILTER_TILE 16        // samples per LDS tile row
#define NBLS_FILTER_GROUP 64       // chunks per carry group
#define NBLS_MAX_PAIRS 512
#define NBLS_LDS_WINDOW 10000     // samples: beyond, the two windows of xcorr_simple_kernel (2 * W * 8 B) do not fit a CU's 160 KB of LDS and are read from global memory
#define NBLS_MAX_STARTS 1024
#define NBLS_MAX_CAND 16

// Per-handle switches (nbls_set_option).  The first group selects between implementations that give
// IDENTICAL results (A/B timing, tests that compare kernels with each other).  The second group exists only
// in a -DNBLS_DEVELOPER build (`make dev`): in-kernel time stamps and ablation switches that make results
// WRONG on purpose; the shipped library has none of that code in its kernels.
struct nbls_options {
    int lts_impl = 0;          // 0 auto; 1 lane-per-start generic LTS kernel everywhere; 3 generic only where no register kernel exists
    int lts_generic_h = 0;     // 1: the register LTS kernel without the h-specialised instantiation
    int lts_coop_threads = 0;  // > 0: workgroup size (64..512) of the large-array LTS kernel (solve_bucket.inc)
    int lts_sample_its = 0;    // large-array LTS kernel: the first n selections of a start take the coarse sample pass (0: all of them, < 0: none)
    int screen_nsl1 = 0;       // 1: one sliding channel per screening workgroup
    int screen_tb4 = 0;        // 1: four-tile lag groups also where the eight-tile instance of the screening kernel applies
    int screen_static = 0;     // 1: fixed (snake-order) deal of the lag groups instead of the dynamic one
    int screen_pretest = 0;    // 1: integer pre-test of a lag group's accumulators before the f32 conversion (epilogue)
    int screen_cxx = 0;        // 1: the compiler-scheduled K loop of the screening kernel everywhere (experiment)
    int screen_nc4 = 0;        // 1: four byte-shifted copies per sliding channel also where eight fit (experiment)
    int screen_tb8 = 0;        // 1: the eight-tile instance wherever one lag block per tile step applies (S == 1)
    int screen_batch_mb = 192;  // quantised-window bytes per unit batch
    int solve_min_units = 0;   // > 0: units a per-batch solve (and a streamed result batch) covers at least (default 8192)
    int filter_row_step = 0;   // > 0: the filter stage runs in launches of at most this many channels even when the whole trace is there
    int result_tail_units = 0; // streamed pass: the LAST result batch is cut to this many units (0: default 2048, < 0: not cut), see xcorr_screen.hip
    int overlap = 0;           // solve of batch k on a second stream while batch k+1 is correlated: 1 on, -1 off, 0 auto (streamed passes of several small batches)
    int filter_nofuse = 0;     // 1: separate state kernel for the backward filter pass
    int filter_nomfma = 0;     // 1: VALU state kernel
    // ---- developer build only ----
    int ablate = 0;            // skip parts of the screening / verify kernels (timing; results wrong)
    int screen_stamps = 0;     // s_memtime phase stamps of the screening kernel
    int screen_seed = 0;       // experiment: running maxima of the screening kernel seeded from the previous pass of the same batch
    int lts_stamps = 0;        // ... of the wave-per-unit LTS kernel
    int screen_pad_kb = 0;     // extra LDS per screening workgroup (occupancy experiment)
    int lts_pad_kb = 0;        // extra LDS per LTS workgroup
    int plan_timing = 0;       // print the host phases of nbls_plan
};

// Consecutive bands of one window length: the unit of the correlator choice (nbls_plan / nbls_launch_xcorr).
struct nbls_wgroup { int b0, b1, W; int64_t u0, u1; bool screen; };

struct nbls_handle {
    int device = 0;
    nbls_options opt;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // solve of batch k runs here while batch k+1 is correlated on `stream`
    hipEvent_t ev_xd = nullptr;        // recorded behind the correlation stage of every pass (nbls_execute_after)
    bool ev_xd_recorded = false;
    hipEvent_t ev_plan = nullptr;      // plan-time uploads wait for this handle's queued kernels (StreamGuard)
    bool ev_xd_by_launcher = false;    // the screening launcher recorded it ahead of its join with the solve stream
    nbls_handle* after = nullptr;      // set for the duration of nbls_execute_after
    int num_cus = 0;                   // compute units of the device (persistent grids)
    hipStream_t up = nullptr;          // plan-time table uploads (highest priority, see alloc_copy)
    int stream_priority = 0;           // nbls_set_option("stream_priority")
    std::vector<hipEvent_t> pev;    // pipeline hand-off events
    bool fuse_solve = false;        // set by nbls_execute_stages: every unit batch of the correlation stage is followed by its solve
    bool solve_on_stream2 = false;  // ... on the second stream (option "overlap"), else behind the batch on `stream`
    bool solve_done = false;
    int last_stage_mask = 7;        // stages of the last pass (nbls_fetch zeroes the outputs of stages that did not run)

    // ---- streamed results (nbls_stream_results): a pinned host mirror of the result block, filled batch by batch ----
    bool stream_results = false;
    unsigned char* h_res = nullptr; // pinned mirror of d_res
    size_t cap_hres = 0;
    hipStream_t cstream = nullptr;  // the D2H copies of the batches (a DMA engine beside the compute streams)
    struct result_batch { int64_t u0, u1, c0, c1; };
    std::vector<result_batch> rbatches;   // batches queued by the last nbls_execute*, in the order they finish
    std::vector<hipEvent_t> rev;    // 2 per batch: [2k] rows complete on the producing stream, [2k+1] copy landed
    std::string err;
    std::mutex err_mu;                 // fail() may be called from the upload thread (nbls_upload_rows) too
    bool trace_loaded = false;         // samples behind the declared shape (nbls_set_trace_shape / nbls_upload_rows)
    // The rows of a trace go up on a stream of their own, an event behind each: a pass queued while nbls_upload_rows is
    // still running on another thread filters the channels as they land (nbls_execute_stages) instead of waiting for the
    // last one — 16 elements x 24 h at 100 Hz are 1.1 GB = 20 ms over PCIe, the filter of a 12-band share 11 ms.
    hipStream_t ustream = nullptr;
    std::vector<hipEvent_t> uev;       // uev[c]: channel c is in HBM
    hipEvent_t ev_uprev = nullptr;     // what was queued on the compute streams before the upload (it may still read d_trace)
    std::atomic<int> rows_landed{0};   // channels whose copy is queued and whose event is recorded (published by the upload thread)
    std::atomic<int> upload_state{0};  // 0 no upload under way, 1 running, 2 shape declared (no samples yet), -1 the upload failed

    // ---- trace (HBM resident) ----
    double* d_trace = nullptr;     // [nchans][npts_pad]
    size_t cap_trace = 0;
    int nchans = 0;
    int64_t npts = 0, npts_pad = 0;
    double fs = 0.0;

    // ---- geometry ----
    int npairs = 0;
    double* d_xij = nullptr;       // [P][2]
    int32_t* d_pair = nullptr;     // [P][2]
    double* d_xpinv = nullptr;     // [2][P]
    std::vector<double> h_xij;
    std::vector<double> h_tl, h_tr;   // host copies of the taper ramps (the same for every band group and call of one trace length: uploaded once)
    std::vector<int32_t> h_pair;   // host copies of the other two geometry tables: an identical nbls_set_geometry uploads nothing
    std::vector<double> h_xpinv;

    // ---- plan ----
    bool planned = false;
    int nbands = 0, nsections = 0, zero_phase = 0, taper_len = 0, vector_len = 0, xcorr_impl = 0;
    std::vector<int32_t> W, inc, nwin, unit_off;
    int64_t nunits = 0;
    int maxW = 0;
    int uniW = 0;                  // the window length if every band has the same one, else 0
    int64_t nchunks = 0;
    double* d_sos = nullptr;       // [B][S][6]
    double* d_M = nullptr;         // [B][G+1][D][D] powers M^0..M^G of the chunk transition (D = 2S)
    double* d_fw = nullptr;        // [B][C][D] zero-state end-state weights
    double* d_gend = nullptr;      // [ngroups][B*N][D]
    double* d_gin = nullptr;       // [ngroups][B*N][D]
    size_t cap_gend = 0, cap_gin = 0;
    double* d_seg_state = nullptr; // [2][B*N][D] initial / final state of a time segment (nbls_filter_segment)
    size_t cap_seg_state = 0;
    double* d_tl = nullptr;        // [taper_len]
    double* d_tr = nullptr;        // [taper_len]
    int32_t* d_W = nullptr;        // [B]
    int32_t* d_inc = nullptr;      // [B]
    int32_t* d_nwin = nullptr;     // [B]
    int32_t* d_unit_off = nullptr; // [B+1]
    int32_t* d_win_off = nullptr;  // [B] first window processed per band
    std::vector<int32_t> win_first, win_count;   // optional per-band window ranges (nbls_set_window_ranges)
    int32_t* d_unit_band = nullptr;// [U]
    int32_t* d_unit_win = nullptr; // [U] window index (global, inside the band) of every unit: saves the kernels a dependent load

    // ---- work + results ----
    double* d_filt = nullptr;      // [B][N][npts_pad]
    double* d_cstate = nullptr;    // [B*N][nchunks][D]
    double* d_cstate2 = nullptr;   // same, for the backward pass (its chunk states are produced by the forward apply)
    size_t cap_cstate2 = 0;
    double* d_tstate = nullptr;    // [B*N][C/T][nchunks][D] forward states at the tile boundaries (zero-phase, recompute form)
    size_t cap_tstate = 0;
    int32_t* d_lag = nullptr;      // [B][VL][P]
    double* d_cmax = nullptr;      // [B][VL][P]
    // result block, ONE allocation = one D2H copy / one RCCL gather:
    //   [vel | baz | mdccm | sigma_tau] double[4][B][VL], then the LTS weight bit mask uint8[B][VL][MB],
    //   MB = ceil(P/8), bit k & 7 of byte k >> 3 = weight of pair k (ceil(P/8) bytes per unit)
    unsigned char* d_res = nullptr;
    size_t cap_res = 0, res_bytes = 0;
    bool res_loaded = false;       // d_res holds a block put there by nbls_load_result_block (cleared by the next nbls_plan)
    size_t reserve_res = 0;        // minimum allocation of the result block (nbls_reserve_results: equal gather blocks)
    // ---- RCCL gather (comm.hip) ----
    void* comm = nullptr;          // ncclComm_t
    int comm_world = 1, comm_rank = 0;
    unsigned char* d_gather = nullptr;   // [world][block_bytes] receive side
    size_t cap_gather = 0;
    int64_t gather_status = 0;     // host copy of the status word while its H2D copy is in flight
    int mask_bytes = 0;            // MB
    double* d_vel = nullptr;       // [B][VL]   (views into d_res)
    double* d_baz = nullptr;
    double* d_mdccm = nullptr;
    double* d_sig = nullptr;
    uint8_t* d_mask = nullptr;     // [B][VL][MB] (view into d_res)
    // confidence intervals of the slowness estimate (nbls_set_uncertainty): computed behind the solve when wanted
    bool want_unc = false;
    double unc_par[6] = {0, 0, 0, 0, 0, 0};   // eigenvalues of X^T X, rotation into the eigen-frame (row major)
    double* d_unc = nullptr;       // [2][B][VL]: vel_uncert | baz_uncert
    size_t cap_unc = 0;
    double* d_z = nullptr;         // [B][VL][2]
    uint8_t* d_wts = nullptr;      // [B][VL][P] one byte per pair (kernel-side form; packed into d_mask after the solve)
    size_t cap_filt = 0, cap_cstate = 0, cap_lag = 0, cap_cmax = 0, cap_z = 0, cap_wts = 0;
    std::unordered_map<const void*, size_t> caps;   // capacities of the small plan tables, keyed by the address of the pointer member
    // pinned staging of the plan tables (api.hip: alloc_copy): a copy from pinned memory goes through the DMA engines,
    // a copy from pageable memory is a shader copy that has to find a free CU — with three other band groups of the
    // call filling the GPU each of a plan's ~25 small uploads took ~60 us instead of ~5
    unsigned char* stage = nullptr;
    size_t stage_cap = 0, stage_used = 0;
    // a plan whose tables all went through the arena does not wait for their copies: the upload stream records ev_up,
    // and whatever launches kernels that read the tables (nbls_execute*, nbls_filter_segment) makes its streams wait
    // for it on the GPU; the next plan / geometry call waits for the upload stream before it reuses the arena
    hipEvent_t ev_up = nullptr;
    bool up_pending = false, stage_bypass = false;
    // nbls_plan's tables live in ONE device arena at the offsets they have in the staging arena and go up in ONE copy
    // when the plan returns (a plan is ~25 tables; each HIP call of a plan that runs beside the trace upload of a
    // pipelined call's first group waited for the runtime's lock: 0.8 instead of 0.4 ms before the first launch)
    std::vector<double> hp_M, hp_FW;       // host side of the plan's filter tables, kept between plans (no allocation per call)
    std::vector<int32_t> hp_ub, hp_uw;     // ... and of the unit -> (band, window) tables
    std::vector<int32_t> woff;     // first computed window of every band of the plan (0 unless window-sharded)
    bool work_queued = false;      // kernels that read the plan / geometry tables may still be queued (set by nbls_execute*, cleared by
                                   // the calls that wait for the handle's stream): a plan has to order its uploads behind them only then
    unsigned char* d_parena = nullptr;
    bool arena_mode = false;
    std::unordered_set<const void*> arena_owned;    // pointer members that point into d_parena (never freed one by one)

    // ---- int8 screening correlator (xcorr_screen.hip) ----
    int8_t* d_qbuf = nullptr;      // [batch][N][2][WP]
    double* d_qmeta = nullptr;     // [batch][N][4]
    int32_t* d_cand = nullptr;     // [batch][N][N][16]
    size_t cap_qbuf = 0, cap_qmeta = 0, cap_cand = 0;
    int64_t screen_batch = 0;
    int screen_wp = 0;             // padded window length the screening buffers are sized for (largest screened group)
    int64_t last_batch = 0;        // units of the last screening batch queued (developer statistics)
    std::vector<nbls_wgroup> wgroups;
    int skew_n = -1, skew_s = -1;  // partner-image skew of the screening kernel, solved once per (array size, tile shape)
    int skew_o[32] = {0};
    int64_t lts_stamp_waves = 0;              // developer: waves of the last LTS launch that wrote stamps
    unsigned long long* d_stamps = nullptr;   // developer: s_memtime stamps of the screen kernel's workgroups
    size_t cap_stamps = 0;

    // ---- LTS ----
    bool lts = false;
    nbls_lts_params ltsp{};
    int32_t* d_starts = nullptr;   // [S][4]
    double* d_rew = nullptr;       // [P+1]
    double* d_xs = nullptr;        // [P][2] standardised co-array
    double* d_xss = nullptr;       // [ceil(P/4)][2] every 4th row of d_xs (padded likewise)
    double* d_xc = nullptr;        // [P] c0*c1 of the standardised co-array; d_xs and d_xc are padded by 16 pairs (solve_bucket.inc reads one block ahead)
    size_t cap_starts = 0;

    // ---- profiling ----
    bool prof = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false;
    std::vector<hipEvent_t> bev;   // per-batch events of the screening path (5 per batch: quantize | screen | verify | solve)
    int bev_used = 0;
    bool prof_fused = false;       // the last profiled pass ran its solves per batch (bev[5k+4] recorded)
    int xcorr_impl_used = 0;       // 1 VALU, 2 f64 MFMA, 3 int8 screening
    nbls_timings tim{};
};

// Kernel launchers (each returns hipError_t of the launch).
hipError_t nbls_launch_filter(nbls_handle* h, int ch0, int nch);    // channels [ch0, ch0 + nch) of every band
hipError_t nbls_launch_filter_segment(nbls_handle* h, int reverse, const double* d_init, double* d_fin);
hipError_t nbls_launch_xcorr(nbls_handle* h);
hipError_t nbls_launch_solve(nbls_handle* h);
hipError_t nbls_launch_solve_range(nbls_handle* h, int64_t u0, int64_t nu, hipStream_t st);
hipError_t nbls_launch_pack_weights(nbls_handle* h, int64_t u0, int64_t nu, hipStream_t st);
// streamed results: queue the copy of the rows of units [u0, u1) into the pinned mirror behind what `producer` has queued
hipError_t nbls_queue_result_batch(nbls_handle* h, int64_t u0, int64_t u1, hipStream_t producer);
hipError_t nbls_launch_probe_mfma(nbls_handle* h, const double* da, const double* db, double* dout);
bool nbls_screen_geometry(const nbls_handle* h, int maxW, int* S, int* PFB, int* CSB, int* CSA, int* WP, size_t* lds, int* nsl, int* G, int* ncopy);
hipError_t nbls_launch_xcorr_screen_range(nbls_handle* h, int64_t ub, int64_t ue, int gW, int64_t* launches_io);
hipError_t nbls_xcorr_screen_finish(nbls_handle* h, int64_t launches);
hipError_t nbls_launch_probe_mfma_i8(nbls_handle* h, const int* da, const int* db, int* dout);
