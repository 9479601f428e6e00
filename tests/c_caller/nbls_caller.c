/* A plain C caller of libnbls_hip.so, compiled against include/nbls.h only (no Python, no ctypes):
 * what a non-Python host binds.  Reads a problem from a flat binary file written by
 * tests/test_c_caller.py, runs it through nbls_run() — plan + execute + sync + fetch in one call — and
 * writes the outputs next to it.
 *
 * File layout (little endian):  int32 nchans, npairs, nbands, nsections, zero_phase, taper_len,
 * vector_len, lts_flag;  int64 npts;  double fs;  then double trace[nchans][npts], xij[npairs][2],
 * int32 pair_idx[npairs][2], double xpinv[2][npairs], sos[nbands][nsections][6], taper_left[taper_len],
 * taper_right[taper_len], int32 winlen[nbands], wininc[nbands];  if lts_flag: double alpha, int32 h, nstarts,
 * csteps, csteps2, ncand, int32 starts[nstarts][4], double xij_mad[2], raw_factor, rew_table[npairs+1],
 * quantile, zero_scale.
 * Output: double vel, baz, mdccm, sigma_tau [nbands][vector_len], int32 nwin[nbands],
 * int32 lag[nbands][vector_len][npairs], uint8 weights[nbands][vector_len][npairs]. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "nbls.h"

/* the rows of a trace uploaded by another thread, a moment after the pass was announced (nbls_expect_upload) */
struct upload_job { nbls_handle* h; const double* const* rows; int32_t nchans; int64_t npts; int rc; };
static void* upload_later(void* p) {
    struct upload_job* j = p;
    const struct timespec ts = {0, 2000000};      /* 2 ms: the other thread is planning, or already waiting in nbls_execute */
    nanosleep(&ts, NULL);
    j->rc = nbls_upload_rows(j->h, j->rows, j->nchans, j->npts);
    return NULL;
}

static void rd(void* p, size_t n, FILE* f) {
    if (n && fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t hd[8];
    int64_t npts;
    double fs;
    rd(hd, sizeof hd, f); rd(&npts, 8, f); rd(&fs, 8, f);
    const int nchans = hd[0], P = hd[1], B = hd[2], S = hd[3], zero_phase = hd[4], tl_n = hd[5], VL = hd[6], lts_flag = hd[7];
    double* trace = malloc((size_t)nchans * npts * 8);
    double* xij = malloc((size_t)P * 2 * 8);
    int32_t* pair = malloc((size_t)P * 2 * 4);
    double* xpinv = malloc((size_t)P * 2 * 8);
    double* sos = malloc((size_t)B * (S ? S : 1) * 6 * 8);
    double* tl = malloc((size_t)(tl_n ? tl_n : 1) * 8);
    double* tr = malloc((size_t)(tl_n ? tl_n : 1) * 8);
    int32_t* winlen = malloc((size_t)B * 4);
    int32_t* wininc = malloc((size_t)B * 4);
    rd(trace, (size_t)nchans * npts * 8, f); rd(xij, (size_t)P * 16, f); rd(pair, (size_t)P * 8, f); rd(xpinv, (size_t)P * 16, f);
    rd(sos, (size_t)B * S * 48, f); rd(tl, (size_t)tl_n * 8, f); rd(tr, (size_t)tl_n * 8, f);
    rd(winlen, (size_t)B * 4, f); rd(wininc, (size_t)B * 4, f);
    nbls_lts_params lp;
    int32_t* starts = NULL;
    double* rew = NULL;
    if (lts_flag) {
        int32_t iv[5];
        rd(&lp.alpha, 8, f); rd(iv, sizeof iv, f);
        lp.h = iv[0]; lp.nstarts = iv[1]; lp.csteps = iv[2]; lp.csteps2 = iv[3]; lp.ncand = iv[4];
        starts = malloc((size_t)lp.nstarts * 16);
        rew = malloc((size_t)(P + 1) * 8);
        rd(starts, (size_t)lp.nstarts * 16, f); rd(lp.xij_mad, 16, f); rd(&lp.raw_factor, 8, f);
        rd(rew, (size_t)(P + 1) * 8, f); rd(&lp.quantile, 8, f); rd(&lp.zero_scale, 8, f);
        lp.starts = starts;
        lp.rew_table = rew;
    }
    fclose(f);

    nbls_handle* h = NULL;
    int rc = nbls_create(0, &h);
    if (rc) { fprintf(stderr, "nbls_create: %d %s\n", rc, nbls_last_error(NULL)); return 1; }
    if ((rc = nbls_set_trace(h, trace, nchans, npts, fs)) || (rc = nbls_set_geometry(h, xij, pair, xpinv, P))) {
        fprintf(stderr, "setup: %d %s\n", rc, nbls_last_error(h));
        return 1;
    }
    const size_t cells = (size_t)B * VL;
    double* grids = calloc(4 * cells, 8);
    int32_t* nwin = calloc(B, 4);
    int32_t* lag = calloc(cells * P, 4);
    uint8_t* wts = calloc(cells * P, 1);
    rc = nbls_run(h, B, S ? sos : NULL, S, zero_phase, tl, tr, tl_n, winlen, wininc, VL, lts_flag ? &lp : NULL, 0,
                  grids, grids + cells, grids + 2 * cells, grids + 3 * cells, nwin, lag, NULL, wts, NULL);
    if (rc) { fprintf(stderr, "nbls_run: %d %s\n", rc, nbls_last_error(h)); return 1; }
    /* error path of the ABI: a vector_len that is too small must be refused with NBLS_ERR_ARG, not crash */
    if (nwin[0] > 1) {
        const int bad = nbls_plan(h, B, S ? sos : NULL, S, zero_phase, tl, tr, tl_n, winlen, wininc, 1, NULL, 0);
        if (bad != NBLS_ERR_ARG) { fprintf(stderr, "expected NBLS_ERR_ARG, got %d\n", bad); return 1; }
    }
    /* the pipelined form from C: band 0 on this handle, the other bands on a second handle of the same GPU (trace
     * copied device-to-device, correlation stage chained behind the first pass), the trace declared first and its
     * samples uploaded after the plan; packed results must equal what nbls_run returned */
    if (B >= 2) {
        nbls_handle* h2 = NULL;
        const double** rows = malloc((size_t)nchans * sizeof *rows);
        for (int c = 0; c < nchans; ++c) rows[c] = trace + (size_t)c * npts;
        if ((rc = nbls_create(0, &h2))) { fprintf(stderr, "second handle: %d\n", rc); return 1; }
        if (nbls_execute_after(h2, h) != NBLS_ERR_STATE) { fprintf(stderr, "execute_after without a plan must be NBLS_ERR_STATE\n"); return 1; }
        if ((rc = nbls_set_trace_shape(h, nchans, npts, fs)) || (rc = nbls_set_geometry(h, xij, pair, xpinv, P)) ||
            (rc = nbls_plan(h, 1, S ? sos : NULL, S, zero_phase, tl, tr, tl_n, winlen, wininc, VL, lts_flag ? &lp : NULL, 0))) {
            fprintf(stderr, "two-step setup: %d %s\n", rc, nbls_last_error(h));
            return 1;
        }
        if (nbls_execute(h) != NBLS_ERR_STATE) { fprintf(stderr, "execute before nbls_upload_rows must be NBLS_ERR_STATE\n"); return 1; }
        if ((rc = nbls_upload_rows(h, rows, nchans, npts)) || (rc = nbls_execute(h))) {
            fprintf(stderr, "upload / execute: %d %s\n", rc, nbls_last_error(h));
            return 1;
        }
        /* the announced form: the pass is planned AND queued while another thread is still uploading the rows — the
         * filter stage takes the channels as they land (same results: checked against nbls_run below) */
        {
            struct upload_job job = {h, rows, nchans, npts, -1};
            pthread_t th;
            if ((rc = nbls_set_trace_shape(h, nchans, npts, fs)) || (rc = nbls_expect_upload(h))) {
                fprintf(stderr, "announce: %d %s\n", rc, nbls_last_error(h));
                return 1;
            }
            if (nbls_expect_upload(h) != NBLS_ERR_STATE) { fprintf(stderr, "a second announcement must be NBLS_ERR_STATE\n"); return 1; }
            if (pthread_create(&th, NULL, upload_later, &job)) { fprintf(stderr, "pthread_create\n"); return 1; }
            rc = nbls_plan(h, 1, S ? sos : NULL, S, zero_phase, tl, tr, tl_n, winlen, wininc, VL, lts_flag ? &lp : NULL, 0);
            if (!rc) rc = nbls_execute(h);
            pthread_join(th, NULL);
            if (rc || job.rc) { fprintf(stderr, "pass on rows in flight: %d / upload %d: %s\n", rc, job.rc, nbls_last_error(h)); return 1; }
        }
        if ((rc = nbls_set_trace_from(h2, h)) || (rc = nbls_set_geometry(h2, xij, pair, xpinv, P)) ||
            (rc = nbls_plan(h2, B - 1, S ? sos + (size_t)S * 6 : NULL, S, zero_phase, tl, tr, tl_n, winlen + 1, wininc + 1, VL,
                            lts_flag ? &lp : NULL, 0)) ||
            (rc = nbls_execute_after(h2, h))) {
            fprintf(stderr, "second pass: %d %s\n", rc, nbls_last_error(h2));
            return 1;
        }
        int64_t lay[4];
        nbls_handle* hh[2] = {h, h2};
        const int nb[2] = {1, B - 1};
        size_t band0 = 0;
        for (int g = 0; g < 2; ++g) {
            if ((rc = nbls_result_layout(hh[g], lay))) { fprintf(stderr, "layout: %d\n", rc); return 1; }
            unsigned char* blk = malloc((size_t)lay[2]);
            if ((rc = nbls_fetch_packed(hh[g], blk, lay[2]))) { fprintf(stderr, "fetch_packed: %d %s\n", rc, nbls_last_error(hh[g])); return 1; }
            const double* pg = (const double*)blk;
            const size_t pc = (size_t)nb[g] * VL;
            for (int q = 0; q < 3; ++q)                  /* vel, baz, mdccm (sigma_tau stays zero under LTS) */
                for (size_t i = 0; i < pc; ++i) {
                    const double x = pg[q * pc + i], y = grids[q * cells + band0 * VL + i];
                    if (!(x == y || (x != x && y != y))) { fprintf(stderr, "pipelined form differs: grid %d band group %d cell %zu\n", q, g, i); return 1; }
                }
            free(blk);
            band0 += (size_t)nb[g];
        }
        nbls_destroy(h2);
        free(rows);
    }
    nbls_destroy(h);
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 2; }
    fwrite(grids, 8, 4 * cells, f); fwrite(nwin, 4, B, f); fwrite(lag, 4, cells * P, f); fwrite(wts, 1, cells * P, f);
    fclose(f);
    printf("C_CALLER_OK version %d units %d\n", nbls_version(), (int)nwin[0]);
    return 0;
}
