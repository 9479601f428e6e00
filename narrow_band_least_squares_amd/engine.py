"""Host driver of the HIP path: turns a stream + band list into one batched device pass
(filter -> xcorr/lag pick -> MdCCM + OLS|LTS for every band x window) and unpacks the result
grids.  Used by ``ltsva``, ``filter_data``, ``narrow_band_least_squares*`` and ``bench.py``.
"""
import os

import numpy as np

from . import planner
from ._hip import Handle
from .stream import start_datenum

_handles = {}


def default_device():
    """NBLS_DEVICE, else LOCAL_RANK (one process per GPU under torch.distributed.run), else 0."""
    for key in ('NBLS_DEVICE', 'LOCAL_RANK'):
        v = os.environ.get(key)
        if v is not None and v != '':
            return int(v)
    return 0


def get_handle(device=None):
    """Per-(process, device) handle, created lazily so that a fork before first use is safe."""
    dev = default_device() if device is None else int(device)
    key = (os.getpid(), dev)
    h = _handles.get(key)
    if h is None:
        h = Handle(dev)
        _handles[key] = h
    return h


def stream_to_array(st):
    """-> (data (N, npts) float64 C-contiguous, fs, start date number)."""
    nchans = len(st)
    if nchans == 0:
        raise ValueError('empty stream')
    npts = len(st[0].data)
    fs = float(st[0].stats.sampling_rate)
    data = np.empty((nchans, npts), dtype=np.float64)
    for i, tr in enumerate(st):
        if len(tr.data) != npts:
            raise ValueError('All traces must have the same number of samples.')
        data[i] = tr.data
    return data, fs, start_datenum(getattr(st[0].stats, 'starttime', 0.0))


def check_elements(nchans, alpha):
    if nchans < 3:
        raise RuntimeError('At least 3 array elements are needed for the least squares estimate.')
    if alpha < 1.0 and nchans < 4:
        raise RuntimeError('At least 4 array elements are needed for least trimmed squares.')
    if not (0.5 <= alpha <= 1.0):
        raise ValueError('ALPHA must be in [0.5, 1.0].')


def window_times(t0_datenum, fs, W, inc, nwin):
    """t[w] = tvec[w*inc + W//2], tvec = start + (arange(npts)/fs)/86400 (matplotlib dates)."""
    idx = np.arange(nwin) * inc + int(W / 2)
    return t0_datenum + (idx / fs) / 86400.0


class BandBatch:
    """Results of one device pass over ``nbands`` bands (arrays are (nbands, vector_len))."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def max_bands_per_pass(nchans, npts):
    """Bands whose filtered traces fit the HBM budget of one pass (NBLS_MAX_FILTERED_GB, default 160 of
    the 288 GB): each band keeps an (N, npts) float64 copy resident for the correlation stage."""
    budget = float(os.environ.get('NBLS_MAX_FILTERED_GB', '160')) * 2.0 ** 30
    return max(1, int(budget // (8.0 * nchans * (npts + 64))))


def process(data, fs, t0_datenum, rij, band_edges, winlens, winover, alpha, filter_type=None,
            filter_order=None, filter_ripple=None, vector_len=None, device=None, xcorr_impl=0,
            want_lag=False, want_cmax=False, want_z=False, prefiltered=False, handle=None,
            upload=True, window_slice=None):
    """Run the hot path for a list of bands on one GPU.

    window_slice=(k, n): process only the k-th of n contiguous window slices of every band (window
    sharding across GPUs); rows outside the slice stay zero, ``nwin``/``t`` describe the whole band.

    data (N, npts) raw traces; band_edges [(fmin, fmax), ...]; winlens [seconds per band].
    prefiltered=True: ``data`` is already filtered/tapered (``ltsva`` entry), one band.
    More bands than fit in HBM at once are processed in consecutive passes."""
    nchans, npts = data.shape
    cap = max_bands_per_pass(nchans, npts)
    if len(band_edges) > cap and not prefiltered:
        if vector_len is None:
            vector_len = max(1, max(planner.window_plan(npts, fs, wl, winover)[2] for wl in winlens))
        parts = []
        for b0 in range(0, len(band_edges), cap):
            parts.append(process(data, fs, t0_datenum, rij, band_edges[b0:b0 + cap], winlens[b0:b0 + cap], winover,
                                 alpha, filter_type, filter_order, filter_ripple, vector_len, device, xcorr_impl,
                                 want_lag, want_cmax, want_z, False, handle, upload and b0 == 0, window_slice))
        first = parts[0]

        def cat(name):
            vals = [getattr(p, name) for p in parts]
            return None if vals[0] is None else np.concatenate(vals, axis=0)
        return BandBatch(vel=cat('vel'), baz=cat('baz'), mdccm=cat('mdccm'), sigma_tau=cat('sigma_tau'),
                         nwin=cat('nwin'), t=cat('t'), weights=cat('weights'), lag=cat('lag'), cmax=cat('cmax'),
                         z=cat('z'), sos=[s for p in parts for s in p.sos], W=cat('W'), inc=cat('inc'),
                         pair_idx=first.pair_idx, xij=first.xij, nchans=nchans, alpha=alpha, handle=first.handle)
    check_elements(nchans, alpha)
    nb = len(band_edges)
    h = handle if handle is not None else get_handle(device)
    if upload:
        h.set_trace(data, fs)
        xij, pair_idx, xpinv = planner.co_array(rij)
        h.set_geometry(xij, pair_idx, xpinv)
    else:
        xij, pair_idx, xpinv = planner.co_array(rij)
    W = np.empty(nb, dtype=np.int32)
    inc = np.empty(nb, dtype=np.int32)
    nwin = np.empty(nb, dtype=np.int64)
    for b in range(nb):
        W[b], inc[b], nwin[b] = planner.window_plan(npts, fs, winlens[b], winover)
    if vector_len is None:
        vector_len = max(1, int(nwin.max()))
    if nwin.max() > vector_len:
        raise ValueError('could not broadcast %d windows into result rows of length %d '
                         '(vector_len too small for this band)' % (int(nwin.max()), vector_len))
    sos_ret = []
    if prefiltered:
        if nb != 1:
            raise ValueError('a pre-filtered pass has exactly one band')
        sos, zero_phase, tl, tr = None, False, None, None
    else:
        applied = []
        zero_phase = None
        for (fmin, fmax) in band_edges:
            sa, zp, sr = planner.design_bandpass(filter_type, fmin, fmax, filter_order, filter_ripple, fs)
            applied.append(sa)
            sos_ret.append(sr)
            zero_phase = zp
        sos = planner.pad_sections(applied)
        tl, tr = planner.taper_ramps(npts)
    lts = planner.lts_plan(xij, alpha) if alpha < 1.0 else None
    if window_slice is not None:
        k, n = window_slice
        first = (nwin * k) // n
        h.set_window_ranges(first, (nwin * (k + 1)) // n - first)
    try:
        h.plan(sos, zero_phase, tl, tr, W, inc, vector_len, lts=lts, xcorr_impl=xcorr_impl)
    finally:
        if window_slice is not None:
            h.set_window_ranges(None)
    h.execute()
    h.sync()
    out = h.fetch(want_lag=want_lag, want_cmax=want_cmax, want_weights=lts is not None, want_z=want_z)
    t = np.zeros((nb, vector_len))
    for b in range(nb):
        t[b, :nwin[b]] = window_times(t0_datenum, fs, int(W[b]), int(inc[b]), int(nwin[b]))
    return BandBatch(vel=out['vel'], baz=out['baz'], mdccm=out['mdccm'], sigma_tau=out['sigma_tau'],
                     nwin=nwin.astype(int), t=t, weights=out['weights'], lag=out['lag'],
                     cmax=out['cmax'], z=out['z'], sos=sos_ret, W=W, inc=inc, pair_idx=pair_idx,
                     xij=xij, nchans=nchans, alpha=alpha, handle=h)


def stdict_from_weights(weights_row, nwin, t_row, pair_idx, nchans, prefix=''):
    """lts_array's dropped-element dictionary for one band: key ``str(t)`` -> 1-based element
    numbers of both members of every zero-weight pair (first members, then second members);
    ``'size'`` -> number of elements.  ``prefix`` is put in front of every time key (the band prefix of
    narrow_band_least_squares.py:114-124).  Vectorised: one pass over the weight grid, the values are
    slices of one array (a 6 h / 48 band run has ~5*10^4 entries)."""
    stdict = {}
    nwin = int(nwin)
    rows, cols = np.nonzero(np.asarray(weights_row[:nwin]) == 0)
    if len(rows):
        pair_idx = np.asarray(pair_idx)
        counts = np.bincount(rows, minlength=nwin)
        starts = np.cumsum(counts) - counts
        local = np.arange(len(rows)) - starts[rows]
        pos1 = 2 * starts[rows] + local
        big = np.empty(2 * len(rows), dtype=(pair_idx[:1, 0] + 1).dtype)
        big[pos1] = pair_idx[cols, 0] + 1
        big[pos1 + counts[rows]] = pair_idx[cols, 1] + 1
        nz = np.nonzero(counts)[0]
        pieces = np.split(big, np.cumsum(2 * counts[nz])[:-1])
        # repr of a Python float is the text str(numpy.float64) gives (shortest round-trip form)
        keys = [prefix + repr(x) for x in np.asarray(t_row, dtype=np.float64)[nz].tolist()]
        stdict = dict(zip(keys, pieces))
    stdict['size'] = nchans
    return stdict
