"""Drop-in mirror of the reference's ``narrow_band_least_squares.py``: same three functions,
same positional signatures and return tuples, but the (frequency band x time window) double
loop and the solver inside it run as ONE batched pass on the GPU.

Reference: narrow_band_least_squares.py:8-127 (serial), :134-218 (``narrow_band_loop``),
:223-323 (``..._parallel``, joblib over bands).  Here the "parallel" variant shards bands over
the GPUs of a node and collects the result blocks with one RCCL gather inside the library
(``dist.py``); with one GPU it equals the serial call.
"""
import os
import threading

import numpy as np
from scipy import signal

from . import dist, engine, planner
from .helpers import get_rij


def _vector_len(WINLEN_list, WINOVER, st):
    """Result row length.  Reference: narrow_band_least_squares.py:41-47 (note: the hop is taken
    in SECONDS there, then divided by Fs — reproduced as is)."""
    max_WINLEN = WINLEN_list[-1]
    sampinc = int((1 - WINOVER) * max_WINLEN)
    npts = len(st[0].data)
    its = np.arange(0, npts, sampinc)
    nits = len(its) - 1
    Fs = st[0].stats.sampling_rate
    return int(nits / Fs)


def _band_edges(freqlist, FREQ_BAND_TYPE, bands):
    """narrow_band_least_squares.py:69-75."""
    step = 2 if FREQ_BAND_TYPE == '2_octave_over' else 1
    return [(freqlist[ii], freqlist[ii + step]) for ii in bands]


def _bt_caution(winlen, fmin, fmax):
    """narrow_band_least_squares.py:83-87."""
    temp_BT = winlen * (fmax - fmin)
    if temp_BT < 5.0:
        print('CAUTION: BT < 5! Band between ' + str(fmin) + ' Hz and ' + str(fmax) + ' Hz has BT = ' + str(temp_BT))


def _band_prefix(band_number):
    """'<band:02d>_' as narrow_band_least_squares.py:114-124 builds it."""
    return str(band_number).zfill(2) + '_'


def _prefix_stdict(stdict, band_number):
    """Keys -> '<band:02d>_<key>', 'size' kept.  narrow_band_least_squares.py:114-124."""
    out = {}
    for key in stdict:
        if key != 'size':
            out[str(band_number).zfill(2) + '_' + key] = stdict[key]
        else:
            out[key] = stdict[key]
    return out


def _run_bands(bands, WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, freqlist, FREQ_BAND_TYPE,
               freq_resp_list, FILTER_TYPE, FILTER_ORDER, FILTER_RIPPLE, vector_len, rij=None, want_keys=True,
               key_prefixes=None):
    """One device pass over the given band indices -> (BandBatch, w rows, h rows).

    Everything on the host that does not need a GPU result — the filter responses (``sosfreqz``,
    narrow_band_least_squares.py:78-80), the BT caution (:83-87) and the text of the ``stdict`` time
    keys — runs while the pass is in flight (``host_overlap`` of ``engine.process``).  The traces are
    uploaded row by row from the stream's own buffers."""
    rows, fs, t0 = engine.stream_rows(st)
    if rij is None:
        rij = get_rij(lat_list, lon_list, len(rows))
    edges = _band_edges(freqlist, FREQ_BAND_TYPE, bands)
    winlens = [WINLEN_list[ii] for ii in bands]
    out_rows = {}

    def host_side(res):
        # (the two response arrays are made here, while the GPU works: 1.5 MB of fresh pages before the first launch
        #  were a tenth of a millisecond on the call's critical path)
        w_rows = out_rows['w'] = np.zeros((len(bands), len(freq_resp_list)), dtype=complex)
        h_rows = out_rows['h'] = np.zeros((len(bands), len(freq_resp_list)), dtype=complex)
        fast = planner.sosfreqz_bands(res.sos, freq_resp_list, fs)      # SciPy's values (bit for bit), all bands at once
        for n, ii in enumerate(bands):
            if fast is None:
                ww, hh = signal.sosfreqz(res.sos[n], freq_resp_list, fs=fs)
            else:
                ww, hh = fast[0], fast[1][n]
            w_rows[n, :] = ww
            h_rows[n, :] = hh
            _bt_caution(WINLEN_list[ii], edges[n][0], edges[n][1])
        if ALPHA < 1.0 and want_keys:
            res.keys = engine.time_key_text(res.t, res.nwin, key_prefixes)
            res.stdict = engine.new_stdict(engine.n_keys(res.keys))
            res.pattern_cache = engine.new_pattern_cache()

    def group_done(res, b0, b1):
        # the dropped-element dictionary of the bands whose rows just landed, while later groups are still running
        if ALPHA < 1.0 and want_keys:
            engine.stdict_from_mask(res.mask[b0:b1], res.nwin[b0:b1], res.pair_idx, res.nchans, res.keys,
                                    into=res.stdict, k0=int(np.sum(res.nwin[:b0])), cache=res.pattern_cache)

    def units_done(res, u0, u1):
        # ... of the unit batch whose rows just landed (a streamed pass), while the GPU works on the next batch
        if ALPHA < 1.0 and want_keys:
            engine.stdict_from_mask(res.mask, res.nwin, res.pair_idx, res.nchans, res.keys, into=res.stdict,
                                    cache=res.pattern_cache, units=(u0, u1))

    res = engine.process(rows, fs, t0, rij, edges, winlens, WINOVER, ALPHA, FILTER_TYPE, FILTER_ORDER,
                         FILTER_RIPPLE, vector_len=vector_len, host_overlap=host_side, group_done=group_done,
                         units_done=units_done)
    if ALPHA < 1.0 and want_keys and 'size' not in res.stdict:
        res.stdict['size'] = res.nchans            # (no band had a window: lts_array's dictionary still names the array size)
    if ALPHA < 1.0 and want_keys:
        # the helper objects of the dictionary (2*10^4 pattern records, 2.7 MB of key text at the benchmark's shape) are
        # taken apart by the NEXT call of this process behind its GPU pass, or at exit — not between the last row's
        # arrival and the return of this call
        engine.release_later(res.__dict__.pop('pattern_cache', None), res.__dict__.pop('keys', None))
    return res, out_rows['w'], out_rows['h']


def narrow_band_least_squares(WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, NBANDS, w, h, freqlist,
                              FREQ_BAND_TYPE, freq_resp_list, FILTER_TYPE, FILTER_ORDER, FILTER_RIPPLE,
                              rij=None):
    """Narrow-band least-squares / LTS array processing of every band in one GPU pass.

    Arguments and the 9-tuple ``(vel_array, baz_array, mdccm_array, t_array, stdict_all,
    sig_tau_array, num_compute_list, w_array, h_array)`` are those of the reference
    (narrow_band_least_squares.py:8-127).  Rows beyond ``num_compute_list[b]`` are zeros (the
    serial reference leaves them uninitialised, its parallel variant zero-fills).  ``ALPHA ==
    1.0`` fills ``sig_tau_array`` and returns ``stdict_all = None``; ``ALPHA < 1.0`` returns the
    merged dropped-element dictionary and leaves ``sig_tau_array`` zero, as the reference does.
    ``rij`` (2, N) km is an extension: it overrides the lat/lon geometry."""
    vector_len = _vector_len(WINLEN_list, WINOVER, st)
    if len(w) != len(freq_resp_list) or len(h) != len(freq_resp_list):
        raise ValueError('could not broadcast filter response of length %d into rows of length %d'
                         % (len(freq_resp_list), len(w)))
    bands = list(range(NBANDS))
    res, w_array, h_array = _run_bands(bands, WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, freqlist,
                                       FREQ_BAND_TYPE, freq_resp_list, FILTER_TYPE, FILTER_ORDER,
                                       FILTER_RIPPLE, vector_len, rij=rij,
                                       key_prefixes=[_band_prefix(ii + 1) for ii in bands])
    num_compute_list = [int(n) for n in res.nwin]
    if ALPHA == 1.0:
        stdict_all = None
        sig_tau_array = res.sigma_tau
    else:
        stdict_all = res.stdict                  # built group by group while the GPU was still working
        sig_tau_array = np.zeros(res.sigma_tau.shape)          # (calloc: no pages touched)
    return (res.vel, res.baz, res.mdccm, res.t, stdict_all, sig_tau_array, num_compute_list,
            w_array, h_array)


def narrow_band_loop(ii, freqlist, FREQ_BAND_TYPE, freq_resp_list, st, FILTER_TYPE, FILTER_ORDER,
                     FILTER_RIPPLE, lat_list, lon_list, WINLEN_list, WINOVER, ALPHA, vector_len, rij=None):
    """One band (the reference's joblib task body, narrow_band_least_squares.py:134-218) ->
    ``(vel, baz, mdccm, t, stdict_times, stdict_elements, sig_tau, num_compute, w, h)`` with the
    vectors zero-padded to ``vector_len``."""
    res, w_rows, h_rows = _run_bands([ii], WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, freqlist,
                                     FREQ_BAND_TYPE, freq_resp_list, FILTER_TYPE, FILTER_ORDER,
                                     FILTER_RIPPLE, vector_len, rij=rij)
    num_compute = np.array(int(res.nwin[0]))
    if ALPHA == 1.0:
        stdict_times = None
        stdict_elements = None
    else:
        sd = res.stdict
        temp_array = np.array(list(sd.items()), dtype=object)
        stdict_times = temp_array[:, 0]
        stdict_elements = temp_array[:, 1]
    return (res.vel[0], res.baz[0], res.mdccm[0], res.t[0], stdict_times, stdict_elements,
            res.sigma_tau[0], num_compute, w_rows[0], h_rows[0])


def narrow_band_least_squares_parallel(WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, NBANDS, w, h,
                                       freqlist, FREQ_BAND_TYPE, freq_resp_list, FILTER_TYPE, FILTER_ORDER,
                                       FILTER_RIPPLE, rij=None):
    """Band-parallel variant (reference: narrow_band_least_squares.py:223-323, joblib over bands).

    The bands are partitioned over the GPUs of the node by cost (fewer bands than GPUs, or
    ``NBLS_SHARD=windows``: every GPU takes all bands but one contiguous slice of each band's windows),
    every GPU runs the whole hot path for its share, and ONE grouped RCCL operation inside the library
    (``nbls_comm_gather``) collects the result blocks — grids and packed LTS weights together — over
    xGMI.  Either one process drives all visible GPUs (default; ``NBLS_DEVICES`` selects them), or one
    process per GPU was started by a launcher that exports ``RANK``/``WORLD_SIZE``/``LOCAL_RANK`` (then
    every rank returns the complete 9-tuple).  With one GPU this is the batched single-GPU call.  Results
    are identical in every form: no value crosses bands.  No PyTorch is involved."""
    group = dist.get_group()
    if group is None or group.world == 1 and os.environ.get('NBLS_FORCE_DIST_PATH') != '1':
        return narrow_band_least_squares(WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, NBANDS, w, h,
                                         freqlist, FREQ_BAND_TYPE, freq_resp_list, FILTER_TYPE, FILTER_ORDER,
                                         FILTER_RIPPLE, rij=rij)
    world = group.world
    vector_len = _vector_len(WINLEN_list, WINOVER, st)
    rows, fs, t0 = engine.stream_rows(st)
    nchans, npts = len(rows), len(rows[0])
    bands = list(range(NBANDS))
    status, failure, prep = 0, None, None
    by_windows = NBANDS < world or os.environ.get('NBLS_SHARD') == 'windows'
    shards = None
    # the trace goes up to every local GPU on helper threads (the copy runs inside the library, GIL released) while
    # this thread designs the filters of all bands; a pass is planned as soon as the design is there and queued when
    # its GPU's copy has landed
    uploads = []                    # (thread, error list) per local handle
    row_pipeline = engine.row_pipeline_for(nchans, npts)      # long traces: the passes are queued while the rows still go up
    try:
        for hd in group.handles:
            hd.set_trace_shape(nchans, npts, fs)
            if row_pipeline and hasattr(hd, 'expect_upload'):
                hd.expect_upload()
            err = []

            def _up(hd=hd, err=err):
                try:
                    hd.upload_rows(rows)
                except BaseException as e:      # noqa: BLE001 - re-raised where the pass is queued
                    err.append(e)
            th = threading.Thread(target=_up, name='nbls-upload')
            th.start()
            uploads.append((th, err))
    except Exception as e:
        status, failure = 1, e
    try:
        if rij is None:
            rij = get_rij(lat_list, lon_list, nchans)
        edges = _band_edges(freqlist, FREQ_BAND_TYPE, bands)
        prep = engine.prepare(nchans, npts, fs, rij, edges, [WINLEN_list[ii] for ii in bands], WINOVER, ALPHA,
                              FILTER_TYPE, FILTER_ORDER, FILTER_RIPPLE, vector_len)
    except Exception as e:          # a rank that cannot even plan still takes part in the gather (status word)
        status, failure = 1, failure or e
    npairs = nchans * (nchans - 1) // 2
    unit_bytes = 32 + (npairs + 7) // 8
    if by_windows:
        nb_block = NBANDS
    else:
        costs = dist.band_costs(npts, fs, [WINLEN_list[b] for b in bands], WINOVER, npairs)
        shards, contiguous = dist.plan_shards(costs, world)
        nb_block = max(1, max(len(sh) for sh in shards))
    block_bytes = (nb_block * vector_len * unit_bytes + 7) // 8 * 8 + 8       # + the status word

    def landed(i):
        th, err = uploads[i]
        th.join()
        if err:
            raise err[0]

    # The shares are contiguous band ranges: every LOCAL pass also STREAMS its rows to the host (nbls_stream_results), and
    # the dropped-element dictionary — one entry per window, ~130 ns each under the GIL: 6.5 ms at the benchmark's shape,
    # more than an eighth of a GPU pass — is built for the local ranks (= band ranges) while the GPUs are still working:
    # all of it when one process drives every GPU, this rank's 1/world of it under a launcher (the other ranks' entries
    # are made after the gather, from the gathered masks).  The grids still arrive through the ONE RCCL gather below.
    use_stream = False
    if status == 0:
        cap = max(1, engine.max_bands_per_pass(nchans, npts))      # filtered bands one pass may keep in HBM
        use_stream = (not by_windows and ALPHA < 1.0 and contiguous
                      and engine.stream_pays(ALPHA, prep.nwin, npairs) and max(len(sh) for sh in shards) <= cap
                      and all(hasattr(hd, 'wait_result_batch') for hd in group.handles))

        def start(i, hd):
            r = group.ranks[i]
            mine = list(range(NBANDS)) if by_windows else shards[r]
            wsl = (r, world) if by_windows else None
            if len(mine) <= cap:
                # (queued while the rows may still be going up: the library filters the channels as they land)
                try:
                    engine.launch(hd, rows, prep, bands=None if by_windows else mine, window_slice=wsl, reserve_bytes=block_bytes,
                                  trace_ready=True, before_execute=None if row_pipeline else (lambda: landed(i)), stream=use_stream)
                except BaseException:
                    landed(i)                   # the upload's own error is the cause, if it has one
                    raise
                landed(i)
                return
            # the rank's share does not fit the HBM budget of one pass (NBLS_MAX_FILTERED_GB): consecutive passes of
            # <= cap bands, each fetched to the host; the assembled block goes back to the GPU for the ONE gather
            MBr = prep.mask_bytes
            grids = np.zeros((4, len(mine), vector_len))
            mask = np.zeros((len(mine), vector_len, MBr), dtype=np.uint8)
            for k0 in range(0, len(mine), cap):
                sub = mine[k0:k0 + cap]
                engine.launch(hd, rows, prep, bands=sub, window_slice=wsl, reserve_bytes=block_bytes, trace_ready=True,
                              before_execute=(lambda: landed(i)) if k0 == 0 else None)
                out = hd.fetch_packed()
                grids[:, k0:k0 + len(sub)] = np.stack((out['vel'], out['baz'], out['mdccm'], out['sigma_tau']))
                mask[k0:k0 + len(sub)] = out['mask']
            hd.load_result_block(np.frombuffer(grids.tobytes() + mask.tobytes(), dtype=np.uint8))
        errs = [e for e in dist.run_on_handles(start, group.handles) if e is not None]
        if errs:
            status, failure = 1, errs[0]
    for th, _ in uploads:           # (a failed rank never reached its join)
        th.join()

    # host work that needs no GPU result, while the passes run
    F = len(freq_resp_list)
    w_array = np.zeros((NBANDS, F), dtype=complex)
    h_array = np.zeros((NBANDS, F), dtype=complex)
    t_array = keys = None
    if status == 0:
        try:
            fast = planner.sosfreqz_bands(prep.sos_ret, freq_resp_list, fs)
            for n, ii in enumerate(bands):
                ww, hh = signal.sosfreqz(prep.sos_ret[n], freq_resp_list, fs=fs) if fast is None else (fast[0], fast[1][n])
                w_array[n, :], h_array[n, :] = ww, hh
                if 0 in group.ranks:
                    _bt_caution(WINLEN_list[ii], edges[n][0], edges[n][1])
            t_array = engine.all_window_times(prep, t0)
            if ALPHA < 1.0:
                keys = engine.time_key_text(t_array, prep.nwin, [_band_prefix(ii + 1) for ii in bands])
        except Exception as e:
            status, failure = 1, e

    stdict_head, stdict_parts, cum, cache = None, {}, None, None
    if use_stream and status == 0:
        # rank order = band order: the lowest local rank's batches first (the later ranks' rows wait in their pinned mirrors
        # meanwhile).  Local ranks 0, 1, … (a prefix of the rank order) write straight into the final dictionary; a local rank
        # behind a remote one fills a dictionary of its own, merged in rank order after the gather
        try:
            MBs = prep.mask_bytes
            cum = np.concatenate(([0], np.cumsum(prep.nwin))).astype(np.int64)
            smask = np.zeros((NBANDS, vector_len, MBs), dtype=np.uint8)
            stdict_head = engine.new_stdict(engine.n_keys(keys))
            cache = engine.new_pattern_cache()
            order = sorted(range(len(group.handles)), key=lambda i: group.ranks[i])
            for n, i in enumerate(order):
                r, hd = group.ranks[i], group.handles[i]
                if r == n:
                    target = stdict_head
                else:
                    target = stdict_parts[r] = engine.new_stdict(int(cum[shards[r][-1] + 1] - cum[shards[r][0]]) if shards[r] else 0)
                stdict_parts.setdefault(r, None)
                sh = shards[r]
                if not sh:
                    continue
                b0, b1 = sh[0], sh[-1] + 1
                for k in range(hd.result_batches()):
                    u0, u1, c0, c1, _, msrc = hd.wait_result_batch(k)
                    if c1 > c0:
                        smask[b0:b1].reshape(-1, MBs)[c0:c1] = msrc[c0:c1]
                    if u1 > u0:
                        engine.stdict_from_mask(smask, prep.nwin, prep.pair_idx, nchans, keys, into=target, cache=cache,
                                                units=(int(cum[b0]) + u0, int(cum[b0]) + u1))
        except Exception as e:      # noqa: BLE001 - reported through the gather's status word like every other local failure
            status, failure, stdict_head, stdict_parts = 1, e, None, {}

    blocks = group.gather(block_bytes, status)          # the ONE collective
    if failure is not None:
        raise failure
    if blocks is None:                                  # this process does not drive the root GPU
        return None
    stat = np.ascontiguousarray(blocks[:, -8:]).view(np.int64).ravel()
    if np.any(stat != 0):
        raise RuntimeError('narrow_band_least_squares_parallel: rank(s) %s failed' % np.nonzero(stat)[0].tolist())

    MB = prep.mask_bytes
    if by_windows:
        # slices are disjoint and rows outside a slice are zero: grids add, masks OR
        grids = np.zeros((4, NBANDS, vector_len))
        mask = np.zeros((NBANDS, vector_len, MB), dtype=np.uint8)
        for r in range(world):
            g, m = engine.split_block(blocks[r], NBANDS, vector_len, MB)
            grids += g
            mask |= m
    else:
        grids = np.zeros((4, NBANDS, vector_len))
        mask = np.zeros((NBANDS, vector_len, MB), dtype=np.uint8)
        for r in range(world):
            g, m = engine.split_block(blocks[r], len(shards[r]), vector_len, MB)
            grids[:, shards[r], :] = g
            mask[shards[r]] = m
    num_compute_list = [int(n) for n in prep.nwin]
    if ALPHA == 1.0:
        stdict_all = None
        sig_tau_array = grids[3]
    else:
        if stdict_head is None:
            stdict_all = engine.stdict_from_mask(mask, prep.nwin, prep.pair_idx, nchans, keys)
        else:
            # rank order = band order = the order of the reference's dictionary: what a local rank streamed is there already,
            # the rest comes from the gathered masks
            stdict_all = stdict_head
            for r in range(world):
                if r not in stdict_parts:
                    if shards[r]:
                        engine.stdict_from_mask(mask, prep.nwin, prep.pair_idx, nchans, keys, into=stdict_all, cache=cache,
                                                units=(int(cum[shards[r][0]]), int(cum[shards[r][-1] + 1])))
                elif stdict_parts[r] is not None:
                    stdict_all.update(stdict_parts[r])
            if 'size' not in stdict_all:
                stdict_all['size'] = nchans
            engine.release_later(cache)
        sig_tau_array = np.zeros((NBANDS, vector_len))
    return (grids[0], grids[1], grids[2], t_array, stdict_all, sig_tau_array, num_compute_list, w_array, h_array)
