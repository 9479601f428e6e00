"""Drop-in mirror of the reference's ``narrow_band_least_squares.py``: same three functions,
same positional signatures and return tuples, but the (frequency band x time window) double
loop and the solver inside it run as ONE batched pass on the GPU.

Reference: narrow_band_least_squares.py:8-127 (serial), :134-218 (``narrow_band_loop``),
:223-323 (``..._parallel``, joblib over bands).  Here the "parallel" variant shards bands over
the GPUs of a node when a ``torch.distributed`` process group is active (one process per GPU)
and gathers the grids once at the end; without a process group it equals the serial call.
"""
import os

import numpy as np
from scipy import signal

from . import dist, engine
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
               freq_resp_list, FILTER_TYPE, FILTER_ORDER, FILTER_RIPPLE, vector_len, rij=None):
    """One device pass over the given band indices -> (BandBatch, w rows, h rows)."""
    data, fs, t0 = engine.stream_to_array(st)
    if rij is None:
        rij = get_rij(lat_list, lon_list, data.shape[0])
    edges = _band_edges(freqlist, FREQ_BAND_TYPE, bands)
    winlens = [WINLEN_list[ii] for ii in bands]
    res = engine.process(data, fs, t0, rij, edges, winlens, WINOVER, ALPHA, FILTER_TYPE, FILTER_ORDER,
                         FILTER_RIPPLE, vector_len=vector_len)
    w_rows = np.zeros((len(bands), len(freq_resp_list)), dtype=complex)
    h_rows = np.zeros((len(bands), len(freq_resp_list)), dtype=complex)
    for n, ii in enumerate(bands):
        ww, hh = signal.sosfreqz(res.sos[n], freq_resp_list, fs=fs)
        w_rows[n, :] = ww
        h_rows[n, :] = hh
        _bt_caution(WINLEN_list[ii], edges[n][0], edges[n][1])
    return res, w_rows, h_rows


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
                                       FILTER_RIPPLE, vector_len, rij=rij)
    num_compute_list = [int(n) for n in res.nwin]
    if ALPHA == 1.0:
        stdict_all = None
        sig_tau_array = res.sigma_tau
    else:
        stdict_all = {}
        for n, ii in enumerate(bands):
            stdict_all.update(engine.stdict_from_weights(res.weights[n], num_compute_list[n], res.t[n], res.pair_idx,
                                                         res.nchans, prefix=_band_prefix(ii + 1)))
        sig_tau_array = np.zeros_like(res.sigma_tau)
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
        sd = engine.stdict_from_weights(res.weights[0], int(num_compute), res.t[0], res.pair_idx, res.nchans)
        temp_array = np.array(list(sd.items()), dtype=object)
        stdict_times = temp_array[:, 0]
        stdict_elements = temp_array[:, 1]
    return (res.vel[0], res.baz[0], res.mdccm[0], res.t[0], stdict_times, stdict_elements,
            res.sigma_tau[0], num_compute, w_rows[0], h_rows[0])


def narrow_band_least_squares_parallel(WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, NBANDS, w, h,
                                       freqlist, FREQ_BAND_TYPE, freq_resp_list, FILTER_TYPE, FILTER_ORDER,
                                       FILTER_RIPPLE, rij=None):
    """Band-parallel variant (reference: narrow_band_least_squares.py:223-323, joblib over bands).

    With an initialised ``torch.distributed`` process group of W ranks (one process per GPU) the
    bands are partitioned over the ranks by cost, each rank runs its bands on its own GPU, and
    one all-gather makes the complete 9-tuple available on every rank.  Without a process group
    this is the single-GPU batched call.  Results are identical either way: no value crosses
    bands."""
    rank, world, backend = dist.dist_info()
    if world == 1 and not (backend is not None and os.environ.get('NBLS_FORCE_DIST_PATH') == '1'):
        return narrow_band_least_squares(WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, NBANDS, w, h,
                                         freqlist, FREQ_BAND_TYPE, freq_resp_list, FILTER_TYPE, FILTER_ORDER,
                                         FILTER_RIPPLE, rij=rij)
    if NBANDS < world or os.environ.get('NBLS_SHARD') == 'windows':
        return _parallel_by_windows(rank, world, WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, NBANDS,
                                    freqlist, FREQ_BAND_TYPE, freq_resp_list, FILTER_TYPE, FILTER_ORDER,
                                    FILTER_RIPPLE, rij)
    vector_len = _vector_len(WINLEN_list, WINOVER, st)
    npts = len(st[0].data)
    fs = float(st[0].stats.sampling_rate)
    nchans = len(st)
    npairs = nchans * (nchans - 1) // 2
    costs = dist.band_costs(npts, fs, [WINLEN_list[b] for b in range(NBANDS)], WINOVER, npairs)
    shards = dist.shard_bands(costs, world)
    mine = shards[rank]
    maxb = max(1, max(len(s) for s in shards))
    F = len(freq_resp_list)
    grids = np.zeros((5, maxb, vector_len))          # vel, baz, mdccm, t, sigma_tau
    nwin = np.zeros(maxb, dtype=np.int64)
    resp = np.zeros((2, maxb, F), dtype=complex)
    wts = np.zeros((maxb, vector_len, npairs), dtype=np.uint8)
    pair_idx = None
    if mine:
        res, w_rows, h_rows = _run_bands(mine, WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, freqlist,
                                         FREQ_BAND_TYPE, freq_resp_list, FILTER_TYPE, FILTER_ORDER,
                                         FILTER_RIPPLE, vector_len, rij=rij)
        n = len(mine)
        grids[0, :n], grids[1, :n], grids[2, :n] = res.vel, res.baz, res.mdccm
        grids[3, :n], grids[4, :n] = res.t, res.sigma_tau
        nwin[:n] = res.nwin
        resp[0, :n], resp[1, :n] = w_rows, h_rows
        if res.weights is not None:
            wts[:n] = res.weights
        pair_idx = res.pair_idx
    if pair_idx is None:
        from .planner import pair_table
        pair_idx = pair_table(nchans)
    dev = engine.default_device()
    all_grids = dist.all_gather_arrays(grids, dev)
    all_nwin = dist.all_gather_arrays(nwin, dev)
    all_resp = dist.all_gather_arrays(np.ascontiguousarray(resp.view(np.float64)), dev)
    all_wts = dist.all_gather_arrays(wts, dev) if ALPHA < 1.0 else None

    vel_array = np.zeros((NBANDS, vector_len))
    baz_array = np.zeros((NBANDS, vector_len))
    mdccm_array = np.zeros((NBANDS, vector_len))
    t_array = np.zeros((NBANDS, vector_len))
    sig_tau_array = np.zeros((NBANDS, vector_len))
    w_array = np.zeros((NBANDS, F), dtype=complex)
    h_array = np.zeros((NBANDS, F), dtype=complex)
    num_compute_list = [0] * NBANDS
    stdict_all = None if ALPHA == 1.0 else {}
    per_band_dict = {}
    for r in range(world):
        g = all_grids[r]
        rr = all_resp[r].view(complex)
        for n, b in enumerate(shards[r]):
            vel_array[b], baz_array[b], mdccm_array[b], t_array[b] = g[0, n], g[1, n], g[2, n], g[3, n]
            if ALPHA == 1.0:
                sig_tau_array[b] = g[4, n]
            num_compute_list[b] = int(all_nwin[r][n])
            w_array[b], h_array[b] = rr[0, n], rr[1, n]
            if ALPHA < 1.0:
                per_band_dict[b] = engine.stdict_from_weights(all_wts[r][n], num_compute_list[b], t_array[b], pair_idx,
                                                              nchans, prefix=_band_prefix(b + 1))
    if ALPHA < 1.0:
        for b in range(NBANDS):
            stdict_all.update(per_band_dict[b])
    return (vel_array, baz_array, mdccm_array, t_array, stdict_all, sig_tau_array, num_compute_list,
            w_array, h_array)


def _parallel_by_windows(rank, world, WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, NBANDS, freqlist,
                         FREQ_BAND_TYPE, freq_resp_list, FILTER_TYPE, FILTER_ORDER, FILTER_RIPPLE, rij):
    """Fewer bands than GPUs (octave bands, example.py's 8 bands on a bigger node, ...): every rank takes
    ALL bands but only its contiguous slice of each band's windows.  The filter runs
    over the whole trace on every rank (zero-phase filtering is not local in time; it is a few per cent of
    the work), correlation and solve only over the slice.  Slices are disjoint and unprocessed rows are
    zero, so the full grids are the sum of the gathered per-rank grids."""
    vector_len = _vector_len(WINLEN_list, WINOVER, st)
    data, fs, t0 = engine.stream_to_array(st)
    nchans = data.shape[0]
    if rij is None:
        rij = get_rij(lat_list, lon_list, nchans)
    bands = list(range(NBANDS))
    edges = _band_edges(freqlist, FREQ_BAND_TYPE, bands)
    res = engine.process(data, fs, t0, rij, edges, [WINLEN_list[ii] for ii in bands], WINOVER, ALPHA, FILTER_TYPE,
                         FILTER_ORDER, FILTER_RIPPLE, vector_len=vector_len, window_slice=(rank, world))
    F = len(freq_resp_list)
    w_array = np.zeros((NBANDS, F), dtype=complex)
    h_array = np.zeros((NBANDS, F), dtype=complex)
    for n, ii in enumerate(bands):
        ww, hh = signal.sosfreqz(res.sos[n], freq_resp_list, fs=fs)
        w_array[n, :], h_array[n, :] = ww, hh
        if rank == 0:
            _bt_caution(WINLEN_list[ii], edges[n][0], edges[n][1])
    dev = engine.default_device()
    grids = np.stack((res.vel, res.baz, res.mdccm, res.sigma_tau))
    total = np.sum(dist.all_gather_arrays(grids, dev), axis=0)
    num_compute_list = [int(n) for n in res.nwin]
    if ALPHA == 1.0:
        stdict_all = None
        sig_tau_array = total[3]
    else:
        wts = np.sum(dist.all_gather_arrays(res.weights, dev), axis=0, dtype=np.uint8)
        # rows outside every slice do not exist (beyond num_compute); inside, exactly one rank wrote 0/1
        stdict_all = {}
        for n, ii in enumerate(bands):
            stdict_all.update(engine.stdict_from_weights(wts[n], num_compute_list[n], res.t[n], res.pair_idx, nchans,
                                                         prefix=_band_prefix(ii + 1)))
        sig_tau_array = np.zeros_like(total[3])
    return (total[0], total[1], total[2], res.t, stdict_all, sig_tau_array, num_compute_list, w_array, h_array)
