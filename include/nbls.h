/*
 * nbls.h — C ABI of libnbls_hip.so: the MI355X (gfx950) narrow-band least-squares /
 * least-trimmed-squares array processor.
 *
 * The reference (amiezzi/narrow_band_least_squares) is pure Python and has no FFI of its
 * own; this ABI is what a ctypes binding for its hot path binds (INTEGRATION.md shows the
 * stub).  Each entry point names the reference interface it replaces (file:line relative
 * to the reference checkout):
 *
 *   nbls_set_trace      <- the `st` argument of narrow_band_least_squares()
 *                          (narrow_band_least_squares.py:8,43) / ltsva() (:91)
 *   nbls_set_geometry   <- lat_list/lon_list -> rij -> co-array inside ltsva
 *                          (helpers.py:239-284; lts_array DataBin/LsBeam, source absent)
 *   nbls_plan           <- the per-band arguments of the band loop
 *                          (narrow_band_least_squares.py:67-91: filter_data() SOS,
 *                          WINLEN_list[ii], WINOVER, ALPHA)
 *   nbls_execute        <- one pass of the band loop body for every planned band:
 *                          helpers.py:124-139 (filter + taper) and ltsva
 *                          (narrow_band_least_squares.py:91 / :183)
 *   nbls_fetch[_packed] <- the result rows written at narrow_band_least_squares.py:104-113
 *                          (vel/baz/mdccm/sigma_tau), the lag vectors tau and the LTS
 *                          weights that become `stdict` (:114-124)
 *   nbls_run            <- convenience: plan + execute + sync + fetch
 *
 * Conventions: plain pointers and sizes; the caller owns every host buffer; the library
 * keeps no host pointer after a call returns; every function returns 0 on success or a
 * negative nbls_status; nbls_last_error() gives the message.  One handle = one GPU = one
 * host thread at a time.  All floating point is IEEE double.
 */
#ifndef NBLS_H
#define NBLS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbls_handle nbls_handle;

typedef enum {
    NBLS_OK = 0,
    NBLS_ERR_ARG = -1,        /* bad shape / NULL pointer / out-of-range parameter   */
    NBLS_ERR_STATE = -2,      /* call order (no trace / geometry / plan yet)          */
    NBLS_ERR_GEOMETRY = -3,   /* < 3 elements, < 4 for LTS, rank-deficient co-array   */
    NBLS_ERR_HIP = -4,        /* HIP runtime failure                                  */
    NBLS_ERR_NOMEM = -5,      /* device allocation failed                             */
    NBLS_ERR_UNSUPPORTED = -6,/* size beyond what the kernels are built for           */
    NBLS_ERR_COMM = -7        /* RCCL not loadable / communicator failure             */
} nbls_status;

/* FAST-LTS parameters (lts_array LTSEstimator / robustbase ltsReg; host computes them). */
typedef struct {
    double alpha;           /* subset fraction in [0.5, 1)                              */
    int32_t h;              /* subset size                                              */
    int32_t nstarts;        /* number of elemental starts (<= 1024)                     */
    const int32_t* starts;  /* [nstarts][4] pair indices, -1 padded                     */
    int32_t csteps;         /* C-steps per start (4)                                    */
    int32_t csteps2;        /* max C-steps in the refinement (100)                      */
    int32_t ncand;          /* candidates refined (10, <= 16)                           */
    double xij_mad[2];      /* 1.4826 * median |xij| per column                         */
    double raw_factor;      /* raw consistency * small-sample correction factor         */
    const double* rew_table;/* [npairs+1] reweighted factor by number of unit weights   */
    double quantile;        /* weight cut-off (qnorm(0.9875))                           */
    double zero_scale;      /* exact-fit threshold (1e-7)                               */
} nbls_lts_params;

/* Per-run timing of the three stages, measured with HIP events on the handle's stream. */
typedef struct {
    double filter_ms;
    double xcorr_ms;
    double solve_ms;
    double total_ms;
    int64_t xcorr_launches;   /* unit batches of the correlation stage                         */
    double quantize_ms;       /* int8 screening path only: sums over the batches               */
    double screen_ms;         /*   the int8-MFMA screening kernel (dominant kernel)            */
    double verify_ms;         /*   the FP64 verification kernel                                */
    int32_t xcorr_impl;       /* correlator actually used: 1 VALU, 2 f64 MFMA, 3 int8 screening */
    int32_t xcorr_fallback_bands; /* xcorr_impl == 3: bands whose window length does not fit the screening
                                   * kernel's LDS even with partner groups (> ~7900 samples) and ran on a general
                                   * correlator; the other bands of the plan were screened                    */
} nbls_timings;

int nbls_version(void);
/* Number of visible HIP devices (0 if the runtime cannot be initialised). */
int nbls_device_count(void);

/* Create a handle on HIP device `device_id` (lazy HIP init happens here, so it is safe to
 * fork before the first call).  *out is NULL on failure; nbls_last_error(NULL) explains. */
int nbls_create(int device_id, nbls_handle** out);
void nbls_destroy(nbls_handle* h);
const char* nbls_last_error(const nbls_handle* h);

/* Raw multichannel trace, channel-major contiguous trace[nchans][npts], copied to HBM and
 * kept resident until the next nbls_set_trace. */
int nbls_set_trace(nbls_handle* h, const double* trace, int32_t nchans, int64_t npts, double fs);

/* Same, from one pointer per channel (rows[c][npts], each contiguous): the traces of an obspy-like
 * Stream are separate arrays, this form uploads them without first packing them into one host block. */
int nbls_set_trace_rows(nbls_handle* h, const double* const* rows, int32_t nchans, int64_t npts, double fs);

/* Same trace as another handle of the SAME device, copied device-to-device (no second trip over PCIe): the
 * band groups of one call run as concurrent passes on several handles of one GPU. */
int nbls_set_trace_from(nbls_handle* h, const nbls_handle* src);

/* Two-step form of nbls_set_trace_rows (the `st` argument, narrow_band_least_squares.py:8,43; the reference copies it
 * per band, helpers.py:124) for a caller that overlaps the host-to-device copy with planning:
 * nbls_set_trace_shape declares the trace (allocation and shape, no samples), after which nbls_set_geometry and
 * nbls_plan may be called; nbls_upload_rows copies the samples (returns when the rows may be reused) and MAY RUN ON
 * ANOTHER THREAD meanwhile — the one exception to "one handle, one thread at a time".  nbls_execute returns
 * NBLS_ERR_STATE until nbls_upload_rows has returned — unless the upload was ANNOUNCED:
 * nbls_expect_upload (after nbls_set_trace_shape, by the thread that goes on to plan and execute) says that
 * nbls_upload_rows is about to run on another thread.  nbls_execute may then be called while the rows are still going
 * up: the rows travel on a stream of their own with an event behind each, and a pass with a filter stage filters the
 * channels as they land ((band, channel) series are independent: identical results) instead of starting after the last
 * one — the copy of a long trace (16 elements x 24 h at 100 Hz: 1.1 GB, 20 ms over PCIe) hides the filter.  It waits on
 * the host for each next row to be queued, and fails with NBLS_ERR_STATE if the upload fails, is aborted
 * (nbls_abort_upload: the announcing side learned that the rows will not come) or does not start within 120 s. */
int nbls_set_trace_shape(nbls_handle* h, int32_t nchans, int64_t npts, double fs);
int nbls_upload_rows(nbls_handle* h, const double* const* rows, int32_t nchans, int64_t npts);
int nbls_expect_upload(nbls_handle* h);
int nbls_abort_upload(nbls_handle* h);

/* Co-array: xij[npairs][2] (km; pair k = (i,j), i<j, lexicographic; xij = r_i - r_j),
 * pair_idx[npairs][2], xpinv[2][npairs] = pseudo-inverse of xij (OLS). */
int nbls_set_geometry(nbls_handle* h, const double* xij, const int32_t* pair_idx,
                      const double* xpinv, int32_t npairs);

/* Describe the bands to process.
 *   sos[nbands][nsections][6]   second-order sections actually applied (a0 == 1);
 *                               nsections == 0 (sos NULL): the trace is already filtered
 *                               (ltsva() on a filtered stream), it is only copied/tapered
 *   zero_phase                  0 causal single pass, 1 forward-backward
 *   taper_left/right[taper_len] whole-trace taper ramps multiplied onto the first/last samples
 *   winlen/wininc[nbands]       window length / hop in samples
 *   vector_len                  row length of the result grids (>= max windows per band)
 *   lts                         NULL -> ordinary least squares
 *   xcorr_impl                  0 auto, 1 plain VALU kernel, 2 f64-MFMA kernel,
 *                               3 int8-MFMA screening + FP64 verification
 * May be called while a pass of this handle is still queued (nbls_execute is asynchronous): the table uploads of
 * the new plan are ordered behind that pass on the GPU, without a host wait.  Fetch that pass's results BEFORE
 * planning again: the layout nbls_fetch* / nbls_result_layout use is the current plan's.
 */
int nbls_plan(nbls_handle* h, int32_t nbands, const double* sos, int32_t nsections,
              int32_t zero_phase, const double* taper_left, const double* taper_right,
              int32_t taper_len, const int32_t* winlen, const int32_t* wininc,
              int32_t vector_len, const nbls_lts_params* lts, int32_t xcorr_impl);

/* Window sharding (fewer bands than GPUs, or traces too long for one GPU's time
 * budget): restrict the NEXT plans to windows [first[b], first[b] + count[b]) of band b (count < 0 =
 * to the end).  The filter still runs over the whole trace (zero-phase filtering is not local in time),
 * correlation and solve only over the slice; result rows keep their global window index, rows outside
 * the slice stay zero, so the slices of several GPUs combine by addition.  nbands <= 0 resets. */
int nbls_set_window_ranges(nbls_handle* h, int32_t nbands, const int32_t* first, const int32_t* count);

/* Launch the whole pass (filter -> xcorr/lag pick -> MdCCM + OLS|LTS) asynchronously on the
 * handle's stream; nbls_sync waits for it. */
int nbls_execute(nbls_handle* h);
/* Same, restricted to a subset of stages: bit 0 filter, bit 1 xcorr/lag pick, bit 2 solve
 * (filter only = the reference's filter_data(), helpers.py:108-141). */
int nbls_execute_stages(nbls_handle* h, int32_t stage_mask);
/* nbls_execute for several handles of ONE GPU whose passes are queued one after the other (the band groups of one
 * call = consecutive iterations of the reference's band loop, narrow_band_least_squares.py:64-124): the correlation stage of `h` starts when the correlation stage `prev` has queued is through (a GPU-side
 * wait, the call itself returns at once); h's filter stage may run beside it.  The passes then finish in the order
 * they were queued, so the caller can work on the first one's results while the later ones are still running —
 * without this the GPU shares itself between the passes and they all finish together at the end.  Results are the
 * same either way.  NBLS_ERR_STATE if `prev` has not queued a pass, NBLS_ERR_ARG for handles of different devices. */
int nbls_execute_after(nbls_handle* h, nbls_handle* prev);
int nbls_sync(nbls_handle* h);

/* Copy results to host.  Any pointer may be NULL to skip it.
 *   vel/baz/mdccm/sigma_tau [nbands][vector_len]   (zeros beyond nwin[b])
 *   nwin[nbands]
 *   lag [nbands][vector_len][npairs] int32: tau = lag / fs  (lag = W - 1 - argmax)
 *   cmax[nbands][vector_len][npairs] normalised cross-correlation maxima
 *   weights[nbands][vector_len][npairs] uint8 (1 = kept, 0 = dropped by LTS; all 1 for OLS)
 *   z[nbands][vector_len][2] slowness estimate (s/km) */
int nbls_fetch(nbls_handle* h, double* vel, double* baz, double* mdccm, double* sigma_tau,
               int32_t* nwin, int32_t* lag, double* cmax, uint8_t* weights, double* z);

/* Confidence intervals of the slowness estimate — the 7th / 8th returns of ltsva (vel_uncert, baz_uncert at
 * narrow_band_least_squares.py:91, example.py:109; Szuberla & Olson 2004 as used by lts_array): half the spread of the
 * trace velocity over the 90 % confidence ellipse of the slowness vector (semi-axes sqrt(chi2_{0.90,2}) sigma_tau /
 * sqrt(lambda_i) along the eigenvectors of X^T X) and half the angle the ellipse subtends at the origin (NaN when the
 * origin is inside).  Computed on the GPU behind every unit's solve when wanted:
 *   nbls_set_uncertainty(h, eig6)   eig6 = {lambda_0, lambda_1 (ascending), R00, R01, R10, R11}: eigenvalues of X^T X
 *                                   and the rotation into its eigen-frame (c = R z); NULL switches it off (default:
 *                                   the band loop discards these two returns).  Read by the next nbls_plan.
 *   nbls_fetch_uncertainty(h, vel_uncert, baz_uncert)   [nbands][vector_len] each (either may be NULL), zeros beyond
 *                                   nwin[b]; waits for the pass like nbls_fetch. */
int nbls_set_uncertainty(nbls_handle* h, const double* eig6);
int nbls_fetch_uncertainty(nbls_handle* h, double* vel_uncert, double* baz_uncert);

/* Copy the filtered+tapered trace of planned band `band` to host: out[nchans][npts]. */
int nbls_fetch_filtered(nbls_handle* h, int32_t band, double* out);

/* ---- time-segmented filtering (a band whose filtered trace does not fit the HBM budget) ----
 * The trace is fed in consecutive time segments (nbls_set_trace per segment, nbls_plan for its length); the IIR state
 * is handed from segment to segment, so the output equals the whole-trace filter (helpers.py:124-139) up to the
 * rounding of the carried states.
 *   nbls_filter_segment(h, reverse, state_in, state_out)
 *       reverse = 0: ONE causal pass over the resident raw segment, forward in time, into the filtered buffer;
 *       reverse = 1: one causal pass over the filtered buffer IN PLACE, backward in time (the second half of a
 *       zero-phase filter: feed the forward outputs back with nbls_set_filtered, last segment first).
 *       state_in / state_out: [nbands][nchans][2 * nsections] DF2T states entering the segment / leaving its last
 *       whole 512-sample chunk (NULL = zero state / not wanted).  A segment that hands a state on must be a multiple
 *       of 512 samples long (every segment but the last in time).  No taper is applied: the caller multiplies the
 *       whole-trace taper in at global sample positions.
 *   nbls_set_filtered(h, band, data[nchans][npts])   write one band of the filtered buffer (the forward outputs of a
 *       segment, before the backward pass). */
int nbls_filter_segment(nbls_handle* h, int32_t reverse, const double* state_in, double* state_out);
int nbls_set_filtered(nbls_handle* h, int32_t band, const double* data);

/* Device pointers of the result grids (for an RCCL gather straight from HBM):
 * ptrs[0..3] = vel, baz, mdccm, sigma_tau (double[nbands][vector_len]); ptrs[4] = nwin (int32).
 * The four grids are the head of the result block (see nbls_result_layout): always contiguous,
 * ptrs[i+1] - ptrs[i] == *bytes_per_grid. */
int nbls_device_results(nbls_handle* h, void** ptrs, int64_t* bytes_per_grid);

/* The result block: ONE device allocation holding, back to back,
 *     double vel[B][VL], baz[B][VL], mdccm[B][VL], sigma_tau[B][VL], uint8 mask[B][VL][MB]
 * MB = ceil(npairs / 8); bit (k & 7) of mask byte (k >> 3) is the LTS weight of pair k (1 = kept; all
 * 1 for OLS; rows beyond nwin[b] are zero).  This is what narrow_band_least_squares() needs back from
 * the GPU (rows written at narrow_band_least_squares.py:104-113 + the weights behind `stdict`, :114-124).
 *   nbls_result_layout: out4 = {cells = B*VL, MB, total bytes, byte offset of the mask}
 *   nbls_fetch_packed : one D2H copy of the whole block into out[nbytes] (nbytes = total bytes) */
int nbls_result_layout(nbls_handle* h, int64_t* out4);
int nbls_fetch_packed(nbls_handle* h, void* out, int64_t nbytes);

/* ---- streamed results: the rows of a pass reach the host batch by batch, while the pass is still running --------
 * Replaces "wait for the whole pass, then build the result rows and the dropped-element dictionary"
 * (narrow_band_least_squares.py:104-124 runs once per band, as each band's ltsva() returns; here a pass covers all
 * bands, and its units are processed in batches of consecutive (band, window) units):
 *   nbls_stream_results(h, 1)     the NEXT nbls_execute* run every unit batch as a complete chain
 *                                 (correlation -> solve -> weight mask) and queue, behind it, a copy of that batch's
 *                                 rows of the result block into a PINNED host mirror of the block owned by the
 *                                 library (same layout as nbls_result_layout / nbls_fetch_packed)
 *   nbls_result_batches(h, &n)    after nbls_execute*: how many batches the pass was cut into (>= 1)
 *   nbls_wait_result_batch(h, k, out4, &block)
 *                                 wait until batch k has landed; out4 = {u0, u1, c0, c1}: the batch holds the units
 *                                 [u0, u1) of the plan (band-major order: all windows of band 0, then band 1, ...) and
 *                                 its rows are the cells [c0, c1) of each grid / of the mask (cell = band * vector_len
 *                                 + window); *block = the mirror.  Batches finish in index order.  Cells outside
 *                                 the batches waited for so far are undefined; the mirror is valid until the handle's
 *                                 next nbls_execute*.  The host may work on batch k (build its dictionary entries)
 *                                 while the GPU runs batch k+1.
 * Results are identical to the unstreamed pass; nbls_fetch* still work afterwards. */
int nbls_stream_results(nbls_handle* h, int32_t on);
int nbls_result_batches(nbls_handle* h, int32_t* nbatches);
int nbls_wait_result_batch(nbls_handle* h, int32_t k, int64_t* out4, const void** host_block);

/* ---- multi-GPU: ONE grouped RCCL operation collects every GPU's result block ----------------------
 * Replaces the joblib fan-out / collection of narrow_band_least_squares_parallel()
 * (narrow_band_least_squares.py:285 and :291-320).  Bands (or window slices) are sharded by the host;
 * every GPU runs nbls_plan/nbls_execute for its share; nbls_comm_gather moves the blocks over xGMI.
 *
 *   nbls_comm_init_all(hs, n)       one process, n handles on n different devices (ncclCommInitAll);
 *                                   handle i becomes rank i of n
 *   nbls_comm_unique_id(id, 128)    rank 0 of a multi-process job creates the communicator id ...
 *   nbls_comm_init_rank(h, id, world, rank)   ... and every process (one GPU each) joins with it; the
 *                                   128 id bytes travel by any side channel (the Python host: a TCP socket)
 *   nbls_reserve_results(h, bytes)  make the NEXT plans allocate the result block with at least
 *                                   `bytes` (= the common block size of the gather: ranks hold
 *                                   different numbers of bands, the gather needs equal blocks)
 *   nbls_comm_gather(hs, n, root, block_bytes, status, host_out, host_bytes)
 *       hs[0..n): the handles this process drives (n = 1 with one process per GPU).  Every rank sends
 *       block_bytes (a multiple of 8, >= its result block + 8): its result block, padding, and in the
 *       last 8 bytes `status` (int64; 0 = this rank's pass succeeded — lets a failed rank still take
 *       part so that nobody hangs; pass the rank's error code otherwise).  root >= 0: gather to that
 *       rank (grouped ncclSend/ncclRecv); root < 0: ncclAllGather.  The process that drives the root
 *       (all-gather: every process) receives host_out[world][block_bytes]; other processes may pass NULL.
 *       The operation is ordered on the handles' streams after their nbls_execute and returns after the
 *       copy to the host has finished.
 *   nbls_comm_destroy(h)            release the communicator (also done by nbls_destroy)
 * RCCL is resolved with dlopen at the first of these calls (librccl.so.1, librccl.so, /opt/rocm/lib/...).
 *   nbls_comm_set_library(path, allow_shared_device)
 *       for rehearsals on a one-GPU box: resolve the ten entry points from `path` instead (the tests' loopback
 *       stand-in, tests/c_caller/loopback_rccl.cpp); allow_shared_device = 1 lets nbls_comm_init_all take several
 *       handles of one device.  Must come before the first comm call (NBLS_ERR_STATE once RCCL has been resolved
 *       from elsewhere); path NULL = RCCL.  The library reads no environment variable.              */
int nbls_comm_set_library(const char* path, int32_t allow_shared_device);
int nbls_comm_init_all(nbls_handle* const* hs, int32_t n);
int nbls_comm_unique_id(void* id, int32_t nbytes);
int nbls_comm_init_rank(nbls_handle* h, const void* id, int32_t world, int32_t rank);
int nbls_reserve_results(nbls_handle* h, int64_t bytes);
int nbls_comm_gather(nbls_handle* const* hs, int32_t n, int32_t root, int64_t block_bytes, int64_t status,
                     void* host_out, int64_t host_bytes);
/* A rank whose share of the bands does not fit the HBM budget of one pass runs it in several passes
 * (nbls_plan / nbls_execute / nbls_fetch_packed per round), assembles its block on the host and puts it back where
 * nbls_comm_gather sends from.  block[nbytes]: the layout of nbls_result_layout for ALL of the rank's bands.
 * The handle needs a new nbls_plan before its next pass. */
int nbls_load_result_block(nbls_handle* h, const void* block, int64_t nbytes);
int nbls_comm_destroy(nbls_handle* h);

/* Per-handle switches, read by the next nbls_plan / nbls_execute.  Every key of the shipped library selects
 * between implementations that give IDENTICAL results (A/B timing; tests that check kernels against each other):
 *   "lts_impl" 0 auto | 1 lane-per-start generic FAST-LTS kernel | 3 generic only where no register kernel exists;
 *   "lts_generic_h", "lts_coop_threads", "lts_sample_its", "screen_tb4", "screen_tb8", "screen_nsl1", "screen_static",
 *   "screen_pretest", "screen_batch_mb", "solve_min_units" (units a per-batch solve / streamed result batch covers at least),
 *   "result_tail_units" (streamed pass: units of the last result batch, cut off the last solve; 0 = default 2048, < 0 = not cut),
 *   "overlap" (the solve of a unit batch on a second stream beside the next batch's correlation: 1 on, -1 off, 0 auto = on
 *   for streamed passes of several small batches), "filter_nofuse", "filter_row_step" (channels per filter launch: what a
 *   pass queued on an announced upload does as the rows land, forced),
 *   "filter_nomfma";
 *   "stream_priority" (applied at once; the handle must be idle): 0 normal, > 0 lower, < 0 higher, clamped to the
 *   device's range — for several handles of one GPU whose passes run side by side.
 * A developer build (make dev, -DNBLS_DEVELOPER; nbls_developer_build() == 1) adds "ablate" (skips kernel parts,
 * results WRONG), "screen_stamps", "lts_stamps", "screen_pad_kb", "lts_pad_kb", "plan_timing"; in the shipped
 * library those keys return NBLS_ERR_UNSUPPORTED and the corresponding code is not in the kernels.  The library
 * reads no environment variable. */
int nbls_set_option(nbls_handle* h, const char* key, int64_t value);
int nbls_developer_build(void);

/* Enable (1) / disable (0) HIP-event timing of the stages; read the last run's timings. */
int nbls_set_profiling(nbls_handle* h, int32_t on);
int nbls_get_timings(nbls_handle* h, nbls_timings* out);

/* plan + execute + sync + fetch in one call. */
int nbls_run(nbls_handle* h, int32_t nbands, const double* sos, int32_t nsections,
             int32_t zero_phase, const double* taper_left, const double* taper_right,
             int32_t taper_len, const int32_t* winlen, const int32_t* wininc,
             int32_t vector_len, const nbls_lts_params* lts, int32_t xcorr_impl,
             double* vel, double* baz, double* mdccm, double* sigma_tau, int32_t* nwin,
             int32_t* lag, double* cmax, uint8_t* weights, double* z);

/* Debug / self-test: run one f64 MFMA 16x16x4 on caller data (a[64], b[64] one value per
 * lane) and return the 256 accumulator values as out[lane*4 + reg]. */
int nbls_probe_mfma_f64(nbls_handle* h, const double* a, const double* b, double* out);
/* Same for one int8 MFMA 16x16x64: a[64][4], b[64][4] packed dwords (16 int8 per lane),
 * out[lane*4 + reg] int32. */
int nbls_probe_mfma_i8(nbls_handle* h, const int32_t* a, const int32_t* b, int32_t* out);

/* Developer statistic of the int8 screening correlator (last unit batch): out8 (EIGHT int64) = {ordered pairs,
 * pairs whose candidate buffer overflowed, total candidates, max candidates per ordered pair, runs of consecutive
 * listed lags, listed lags that sit in a run of two or more, 0, 0}. */
int nbls_debug_screen_stats(nbls_handle* h, int64_t* out8);
/* Developer: mean s_memtime cycle counts of the phases of the wave-per-unit FAST-LTS kernel
(developer build, options "screen_stamps" / "lts_stamps"): out8 = {setup+medians, elemental starts,
 * C-steps, candidate peel, refinement, finish, 0, total}. */
int nbls_debug_lts_stamps(nbls_handle* h, double* out8);
/* Developer: C-step phases of the large-array FAST-LTS kernel (9..32 elements): out8 = mean cycles of thread 0 in
 * {groups (selection + sums), subset merging, compaction}, the live entries summed over the iterations, and wave 0's
 * {selection passes, selection cycles, sums cycles, groups taken}. */
int nbls_debug_lts_coop_breakdown(nbls_handle* h, double* out8);
int nbls_debug_screen_stamps(nbls_handle* h, double* out10);

#ifdef __cplusplus
}
#endif
#endif /* NBLS_H */
