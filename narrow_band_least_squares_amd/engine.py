"""Host driver of the HIP path: turns a stream + band list into one batched device pass
(filter -> xcorr/lag pick -> MdCCM + OLS|LTS for every band x window) and unpacks the result
grids.  Used by ``ltsva``, ``filter_data``, ``narrow_band_least_squares*`` and ``bench.py``.
"""
import itertools
import operator
import os
import threading

import numpy as np

from . import planner
from ._hip import Handle
from .stream import start_datenum

try:                                   # C++ helpers for the stdict keys / dictionary (csrc/host_ext.cpp)
    from . import _nbls_host as _hostext
except ImportError:                    # not built: the pure-Python equivalents below are used
    _hostext = None

_handles = {}

# Scheduling switches of the band-group path of rounds 2-3 and of the host helpers: module attributes, set by tests and
# tools (they were environment variables until round 4: NBLS_UPLOAD_OVERLAP, NBLS_GROUP_ORDER, NBLS_STREAM_PRIORITY,
# NBLS_PIPELINE_SPLIT, NBLS_KEY_THREADS).  Results never depend on them.
UPLOAD_OVERLAP = True      # the trace goes up on a helper thread beside the filter design and the plan
ROW_PIPELINE = True        # ... and the pass is queued while it still does: the library filters the channels as they land
ROW_PIPELINE_MIN_BYTES = 256 << 20   # ... for traces whose upload is worth hiding (1.1 GB at cfg-4: the 12-band share's call 236 ->
                           # 221 ms; at cfg-3's 55 MB = 1.2 ms the extra filter launches cost 0.4 ms more than they hide)
GROUP_ORDER = True         # band groups: the groups' correlation stages are chained on the GPU (nbls_execute_after)
STREAM_PRIORITY = True     # band groups: earlier groups on higher-priority streams
PIPELINE_SPLIT = None      # band groups: explicit shares, e.g. (0.15, 0.5, 0.35)
KEY_THREADS = None         # threads that format the stdict key text (default: min(4, cores - 1))


def default_device():
    """NBLS_DEVICE, else LOCAL_RANK (one process per GPU under torch.distributed.run), else 0."""
    for key in ('NBLS_DEVICE', 'LOCAL_RANK'):
        v = os.environ.get(key)
        if v is not None and v != '':
            return int(v)
    return 0


def get_handle(device=None, slot=0):
    """Per-(process, device, slot) handle, created lazily so that a fork before first use is safe.  Slots > 0
    are further handles (own stream, own buffers) on the same GPU: the band groups of a pipelined call."""
    dev = default_device() if device is None else int(device)
    key = (os.getpid(), dev, int(slot))
    h = _handles.get(key)
    if h is None:
        h = Handle(dev)
        # the groups' passes run side by side; the earlier group's workgroups are dispatched first so that its rows
        # land while the later groups still keep the GPU busy (the host builds that group's dictionary meanwhile)
        if STREAM_PRIORITY:
            h.set_option('stream_priority', min(int(slot), 2) - 1)
        _handles[key] = h
    return h


def pipeline_groups(nwin, npairs=28):
    """Into how many band groups a call is cut (``NBLS_PIPELINE_GROUPS`` overrides).  Each group is its own
    asynchronous pass on its own handle of the same GPU: while the GPU works on group k, the host designs the
    filters of group k+1, and later turns the finished groups' weights into the dropped-element dictionary
    while the remaining groups are still being computed.  Small calls are one group.  "Small" is measured in units
    weighted by the pair count (a unit of a 32-element array costs ~20x the GPU time of an 8-element one, and its
    dictionary entry — up to 62 mask bytes, a value array of its own — several times the host time: at 32 elements x
    128 bands the dictionary of a single-group call was a 15-25 ms tail behind the GPU)."""
    env = os.environ.get('NBLS_PIPELINE_GROUPS')
    nb = len(nwin)
    if env:
        return max(1, min(nb, int(env)))
    weight = max(1.0, float(npairs) / 28.0)
    return max(1, min(4, nb, int(np.sum(nwin) * weight) // 16000))


_graveyard = []


def release_later(*objs):
    """Keep helper objects of a finished call alive until ``release_deferred`` (their teardown — tens of thousands of
    small records — is host time that can hide behind the next call's GPU pass)."""
    _graveyard.append(objs)


def release_deferred():
    del _graveyard[:]


STREAM_MIN_UNITS = 16000    # (band, window) units x (pairs / 28) from which a call streams its rows (``stream_pays``)


def streamed_default():
    """Whether ``NBLS_STREAM_RESULTS`` / ``NBLS_PIPELINE_GROUPS`` allow the streamed form at all (see ``stream_pays``)."""
    if os.environ.get('NBLS_STREAM_RESULTS', '1') == '0':
        return False
    env = os.environ.get('NBLS_PIPELINE_GROUPS')
    return not (env and int(env) > 1)


def stream_pays(alpha, nwin, npairs=28):
    """Whether a whole call runs as ONE pass whose unit batches stream their rows to the host (``nbls_stream_results``)
    while the GPU works on the next batch.  What the host does with a batch's rows is the dropped-element dictionary, so:
    under LTS, from ``STREAM_MIN_UNITS`` units on (weighted by the pair count as in ``pipeline_groups``).  Below that, and
    under OLS (no dictionary), the pass is fetched in one piece: the per-batch solves, copies and events of the streamed
    form are pure overhead there (cfg-1b, 443 OLS units in 8 window groups: 2.2 ms per call streamed, 1.07 ms in one piece;
    cfg-2, 5 700 LTS units: no difference; cfg-3: 17.1 against 18.5 ms).  ``NBLS_STREAM_RESULTS=1`` streams always, ``=0``
    never; ``NBLS_PIPELINE_GROUPS`` > 1 selects the band groups of rounds 2-3."""
    if not streamed_default():
        return False
    if os.environ.get('NBLS_STREAM_RESULTS') == '1':
        return True
    weight = max(1.0, float(npairs) / 28.0)
    return float(alpha) < 1.0 and int(np.sum(nwin) * weight) >= STREAM_MIN_UNITS


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


def stream_rows(st):
    """-> (list of per-channel 1-D float64 arrays — the traces' own buffers when they already are
    C-contiguous float64 —, fs, start date number).  Nothing is packed: ``Handle.set_trace_rows``
    uploads every row from where it lies."""
    nchans = len(st)
    if nchans == 0:
        raise ValueError('empty stream')
    npts = len(st[0].data)
    fs = float(st[0].stats.sampling_rate)
    rows = []
    for tr in st:
        d = np.asarray(tr.data)
        if len(d) != npts:
            raise ValueError('All traces must have the same number of samples.')
        if d.dtype != np.float64 or not d.flags.c_contiguous:
            d = np.ascontiguousarray(d, dtype=np.float64)
        rows.append(d)
    return rows, fs, start_datenum(getattr(st[0].stats, 'starttime', 0.0))


class _UploadWorker:
    """One helper thread per process, kept between calls, that runs the trace uploads of ``process`` (the copy happens
    inside the library with the GIL released).  ``submit(fn)`` -> an object with ``join()``."""

    class _Job:
        def __init__(self, fn):
            self.fn, self.done = fn, threading.Event()

        def join(self):
            self.done.wait()

    def __init__(self):
        import queue
        self.pid = os.getpid()
        self.q = queue.SimpleQueue()
        self.thread = threading.Thread(target=self._run, name='nbls-upload', daemon=True)
        self.thread.start()

    def _run(self):
        while True:
            job = self.q.get()
            try:
                job.fn()                          # (the uploads catch their own exceptions and hand them to the caller)
            finally:
                job.done.set()

    def submit(self, fn):
        job = self._Job(fn)
        self.q.put(job)
        return job


_upload_worker_obj = None
_upload_worker_lock = threading.Lock()


def _upload_worker():
    global _upload_worker_obj
    w = _upload_worker_obj
    if w is None or w.pid != os.getpid() or not w.thread.is_alive():      # (a forked child starts its own)
        with _upload_worker_lock:
            w = _upload_worker_obj
            if w is None or w.pid != os.getpid() or not w.thread.is_alive():
                w = _upload_worker_obj = _UploadWorker()
    return w


def row_pipeline_for(nchans, npts):
    """Queue the pass while the trace is still going up (``Handle.expect_upload``)?  Worth it for long uploads only."""
    return bool(ROW_PIPELINE) and 8 * int(nchans) * int(npts) >= ROW_PIPELINE_MIN_BYTES


def _trace_key(data, fs):
    """Identity of a trace as the caller holds it: where its samples lie (address, length of every row) and the
    sampling rate.  None for anything that is not float64 C-contiguous (such rows are converted per call: no identity)."""
    rows = [data] if isinstance(data, np.ndarray) and data.ndim == 2 else list(data)
    key = [float(fs)]
    for r in rows:
        if not isinstance(r, np.ndarray) or r.dtype != np.float64 or not r.flags.c_contiguous:
            return None
        key.append((r.ctypes.data, r.shape))
    return tuple(key)


class resident_trace:
    """``with engine.resident_trace(st):`` — upload the stream's samples ONCE and keep them in HBM: every call made inside
    the block on the SAME buffers (``narrow_band_least_squares(..., st, ...)``, ``ltsva`` — the same Stream object, its
    traces' ``data`` arrays float64 and C-contiguous, which is what ``stream_rows`` passes through untouched) skips its
    upload (55 MB over PCIe = 1.4 ms of a 17 ms call at the benchmark's shape).  For a caller that runs several
    parameter sets over one trace.  The caller promises not to write to the samples inside the block; any other trace
    processed on the same GPU meanwhile ends the residency (the next call uploads again).  Accepts a Stream, a 2-D array
    or a list of rows (then ``fs`` is required)."""

    def __init__(self, st, fs=None, device=None):
        if fs is None:
            self.rows, self.fs, _ = stream_rows(st)
        else:
            self.rows, self.fs = st, float(fs)
        self.device = device
        self.handle = None

    def __enter__(self):
        key = _trace_key(self.rows, self.fs)
        if key is None:
            raise ValueError('resident_trace: the samples must be float64 and C-contiguous (they are converted per call otherwise)')
        h = get_handle(self.device, 0)
        upload_trace(h, self.rows, self.fs)
        h.resident_key = key
        self.handle = h
        return self

    def __exit__(self, *exc):
        if self.handle is not None and self.handle.resident_key is not None:
            self.handle.resident_key = None
        return False


def _shape_of(data):
    """(nchans, npts) of a 2-D array or of a list of equally long rows."""
    if isinstance(data, np.ndarray):
        if data.ndim != 2:
            raise ValueError('trace must be (nchans, npts)')
        return data.shape
    return len(data), len(data[0])


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
    """Results of one device pass over ``nbands`` bands (arrays are (nbands, vector_len)).
    ``mask`` (nbands, vector_len, ceil(P/8)) is the packed LTS weight mask as the GPU returns it;
    ``weights`` (nbands, vector_len, P) uint8 is unpacked from it on first use."""

    def __init__(self, **kw):
        self._weights = None
        self.__dict__.update(kw)

    @property
    def weights(self):
        if self._weights is None and getattr(self, 'mask', None) is not None and self.lts:
            P = len(self.pair_idx)
            self._weights = np.unpackbits(self.mask, axis=-1, bitorder='little')[..., :P]
        return self._weights


def _filtered_budget_bytes():
    return float(os.environ.get('NBLS_MAX_FILTERED_GB', '160')) * 2.0 ** 30


def max_bands_per_pass(nchans, npts):
    """Bands whose filtered traces fit the HBM budget of one pass (NBLS_MAX_FILTERED_GB, default 160 of
    the 288 GB): each band keeps an (N, npts) float64 copy resident for the correlation stage.  0: not even
    one band fits -> ``process`` switches to the time-segmented path."""
    return int(_filtered_budget_bytes() // (8.0 * nchans * (npts + 64)))


def filter_band_segmented(h, rows, fs, sos_apply, zero_phase, seg_len):
    """Band-pass ONE band of a trace that is too long for the HBM budget: the trace goes through the GPU in
    consecutive time segments of ``seg_len`` samples (a multiple of the 512-sample scan chunk) and the IIR state
    is handed from segment to segment (``nbls_filter_segment``), forward in time and — zero-phase — backward in
    time over the forward outputs.  Equals the whole-trace filter (helpers.py:124-139) up to the rounding of the
    carried states.  -> (N, npts) filtered, UNTAPERED traces on the host."""
    nchans, npts = len(rows), len(rows[0])
    chunk = 512
    seg = max(chunk, int(seg_len) // chunk * chunk)
    bounds = [(a, min(a + seg, npts)) for a in range(0, npts, seg)]
    y = np.empty((nchans, npts))
    sos3 = np.ascontiguousarray(sos_apply, dtype=np.float64)[None, :, :]

    def plan_for(n):                # filter-only plan of an n-sample segment (one dummy window, no geometry needed)
        h.plan(sos3, zero_phase, None, None, [2], [max(1, n)], 1)

    state = None
    for (a, b) in bounds:
        h.set_trace_rows([r[a:b] for r in rows], fs)
        plan_for(b - a)
        state = h.filter_segment(False, state_in=state, want_state=(b < npts))
        y[:, a:b] = h.fetch_filtered(0)
    if zero_phase:
        state = None
        for (a, b) in reversed(bounds):
            seg_y = np.ascontiguousarray(y[:, a:b])
            h.set_trace(seg_y, fs)                          # (sets the segment length; the backward pass reads the filtered buffer)
            plan_for(b - a)
            h.set_filtered(0, seg_y)
            state = h.filter_segment(True, state_in=state, want_state=(a > 0))
            y[:, a:b] = h.fetch_filtered(0)
    return y


def process_segmented(data, fs, t0_datenum, rij, band_edges, winlens, winover, alpha, filter_type, filter_order,
                      filter_ripple, vector_len, device=None, xcorr_impl=0, want_lag=False, want_cmax=False, want_z=False,
                      host_overlap=None, group_done=None):
    """The hot path when not even ONE band's filtered trace fits the HBM budget: band by band,
    (1) the band is filtered in time segments with IIR state hand-off (``filter_band_segmented``) and tapered at
    global positions, (2) its windows go through the correlation + solve kernels in slices of consecutive windows
    (the ``ltsva`` entry of the device pass).  Same rows as the in-core pass up to the rounding of the carried filter
    states; HBM holds one segment / one window slice at a time."""
    rows = [np.ascontiguousarray(r, dtype=np.float64) for r in data]
    nchans, npts = len(rows), len(rows[0])
    nb = len(band_edges)
    h = get_handle(device)
    budget = _filtered_budget_bytes()
    seg_len = max(512, int(budget // (16.0 * nchans)))               # raw + filtered copy of a segment
    W, inc, nwin = [np.empty(nb, dtype=t) for t in (np.int32, np.int32, np.int64)]
    for b in range(nb):
        W[b], inc[b], nwin[b] = planner.window_plan(npts, fs, winlens[b], winover)
    xij, pair_idx, _ = planner.co_array(rij)
    P = xij.shape[0]
    MB = (P + 7) // 8
    grids = np.zeros((4, nb, vector_len))
    mask = np.zeros((nb, vector_len, MB), dtype=np.uint8)
    lag = np.zeros((nb, vector_len, P), dtype=np.int32) if want_lag else None
    cmax = np.zeros((nb, vector_len, P)) if want_cmax else None
    z = np.zeros((nb, vector_len, 2)) if want_z else None
    tt = np.zeros((nb, vector_len))
    for b in range(nb):
        tt[b, :nwin[b]] = window_times(t0_datenum, fs, int(W[b]), int(inc[b]), int(nwin[b]))
    designs = planner.design_bandpass_many(filter_type, band_edges, filter_order, filter_ripple, fs)
    res = BandBatch(vel=grids[0], baz=grids[1], mdccm=grids[2], sigma_tau=grids[3], nwin=nwin.astype(int), t=tt,
                    mask=mask, lag=lag, cmax=cmax, z=z, sos=[d[2] for d in designs], W=W, inc=inc, pair_idx=pair_idx,
                    xij=xij, nchans=nchans, alpha=alpha, handle=h, lts=alpha < 1.0, fs=fs)
    if host_overlap is not None:
        host_overlap(res)
    tl, tr = planner.taper_ramps(npts)
    for b in range(nb):
        sos_apply, zero_phase, _ = designs[b]
        y = filter_band_segmented(h, rows, fs, sos_apply, zero_phase, seg_len)
        if len(tl):
            y[:, :len(tl)] *= tl
            y[:, npts - len(tr):] *= tr
        # windows in slices of consecutive windows: slice [w0, w1) needs samples [w0*inc, (w1-1)*inc + W + 1)
        Wb, ib, nw = int(W[b]), int(inc[b]), int(nwin[b])
        per_slice = max(1, int((budget // (16.0 * nchans) - Wb - 1) // ib))
        for w0 in range(0, nw, per_slice):
            w1 = min(nw, w0 + per_slice)
            s0 = w0 * ib
            L = min(npts - s0, (w1 - w0 - 1) * ib + Wb + 1)
            part = process(np.ascontiguousarray(y[:, s0:s0 + L]), fs, 0.0, rij, [(None, None)], [winlens[b]], winover, alpha,
                           prefiltered=True, handle=h, xcorr_impl=xcorr_impl, want_lag=want_lag, want_cmax=want_cmax,
                           want_z=want_z, vector_len=w1 - w0)
            assert int(part.nwin[0]) == w1 - w0
            grids[:, b, w0:w1] = np.stack((part.vel[0], part.baz[0], part.mdccm[0], part.sigma_tau[0]))
            mask[b, w0:w1] = part.mask[0]
            for name, arr in (('lag', lag), ('cmax', cmax), ('z', z)):
                if arr is not None:
                    arr[b, w0:w1] = getattr(part, name)[0]
        if group_done is not None:
            group_done(res, b, b + 1)
    return res


class Prep:
    """Host-side plan of a call: everything the GPU pass needs that is computed on the host — window
    plan, filter design, taper ramps, co-array, FAST-LTS constants — for ALL bands of the call.  A device
    then runs any subset of the bands (band sharding) or of the windows (window sharding) of it."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def prepare(nchans, npts, fs, rij, band_edges, winlens, winover, alpha, filter_type=None, filter_order=None,
            filter_ripple=None, vector_len=None, prefiltered=False, common=None):
    """``common``: the Prep of another band group of the same call — its co-array, taper ramps and FAST-LTS plan
    do not depend on the bands and are reused instead of being recomputed per group."""
    check_elements(nchans, alpha)
    nb = len(band_edges)
    if common is not None:
        xij, pair_idx, xpinv = common.xij, common.pair_idx, common.xpinv
    else:
        xij, pair_idx, xpinv = planner.co_array(rij)
    W = np.empty(nb, dtype=np.int32)
    inc = np.empty(nb, dtype=np.int32)
    nwin = np.empty(nb, dtype=np.int64)
    plans = {}                                   # (bands usually share a few window lengths: one plan per length)
    for b in range(nb):
        wl = float(winlens[b])
        if wl not in plans:
            plans[wl] = planner.window_plan(npts, fs, winlens[b], winover)
        W[b], inc[b], nwin[b] = plans[wl]
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
        # all bands of the group designed in one vectorised pass (bit-identical to SciPy's per-band design)
        for sa, zp, sr in planner.design_bandpass_many(filter_type, band_edges, filter_order, filter_ripple, fs):
            applied.append(sa)
            sos_ret.append(sr)
            zero_phase = zp
        sos = planner.pad_sections(applied)
        tl, tr = (common.tl, common.tr) if common is not None else planner.taper_ramps(npts)
    if common is not None:
        lts = common.lts
    else:
        lts = planner.lts_plan(xij, alpha) if alpha < 1.0 else None
    return Prep(nchans=nchans, npts=npts, fs=fs, nbands=nb, xij=xij, pair_idx=pair_idx, xpinv=xpinv, W=W, inc=inc,
                nwin=nwin, vector_len=int(vector_len), sos=sos, zero_phase=zero_phase, tl=tl, tr=tr, lts=lts,
                sos_ret=sos_ret, alpha=alpha, npairs=xij.shape[0], mask_bytes=(xij.shape[0] + 7) // 8)


def upload_trace(h, data, fs):
    if isinstance(data, np.ndarray):
        h.set_trace(data, fs)
    else:
        h.set_trace_rows(data, fs)


def launch(h, data, prep, bands=None, upload=True, window_slice=None, xcorr_impl=0, reserve_bytes=0, trace_from=None,
           trace_ready=False, after=None, before_execute=None, stream=False, uncert=False):
    """Upload (optional), plan and start the pass for the band subset ``bands`` (indices into the Prep;
    None = all) on handle ``h``.  Returns as soon as the kernels are queued.  ``trace_from``: another handle of
    the same GPU that already holds this trace (device-to-device copy instead of a second upload).
    ``trace_ready``: the caller has already uploaded the trace to ``h`` (``upload_trace``).  ``after``: the handle
    of the band group queued before this one (``Handle.execute``).  ``before_execute()``: called between plan and
    execute (the caller joins its upload thread there)."""
    if upload:
        if trace_ready:
            pass
        elif trace_from is not None and trace_from is not h:
            h.set_trace_from(trace_from)
        else:
            upload_trace(h, data, prep.fs)
        h.set_geometry(prep.xij, prep.pair_idx, prep.xpinv)
    idx = np.arange(prep.nbands) if bands is None else np.asarray(bands, dtype=np.int64)
    sos = None if prep.sos is None else prep.sos[idx]
    if window_slice is not None:
        k, n = window_slice
        nw = prep.nwin[idx]
        first = (nw * k) // n
        h.set_window_ranges(first, (nw * (k + 1)) // n - first)
    h.reserve_results(reserve_bytes)
    h.set_uncertainty(planner.uncertainty_frame(prep.xij) if uncert else None)     # (ltsva's confidence intervals, on request)
    try:
        h.plan(sos, prep.zero_phase, prep.tl, prep.tr, prep.W[idx], prep.inc[idx], prep.vector_len, lts=prep.lts,
               xcorr_impl=xcorr_impl)
    finally:
        if window_slice is not None:
            h.set_window_ranges(None)
    if before_execute is not None:
        before_execute()
    h.stream_results(stream)             # (``stream``: the rows come back batch by batch, see ``process``)
    h.execute(after=after)


def all_window_times(prep, t0_datenum):
    t = np.zeros((prep.nbands, prep.vector_len))
    for b in range(prep.nbands):
        t[b, :prep.nwin[b]] = window_times(t0_datenum, prep.fs, int(prep.W[b]), int(prep.inc[b]), int(prep.nwin[b]))
    return t


def split_block(block, nbands, vector_len, mask_bytes):
    """A result block (bytes as the GPU wrote them, see nbls_result_layout) -> (grids (4, B, VL) float64,
    mask (B, VL, MB) uint8) views."""
    cells = nbands * vector_len
    grids = block[:32 * cells].view(np.float64).reshape(4, nbands, vector_len)
    mask = block[32 * cells:32 * cells + cells * mask_bytes].reshape(nbands, vector_len, mask_bytes)
    return grids, mask


def process(data, fs, t0_datenum, rij, band_edges, winlens, winover, alpha, filter_type=None,
            filter_order=None, filter_ripple=None, vector_len=None, device=None, xcorr_impl=0,
            want_lag=False, want_cmax=False, want_z=False, prefiltered=False, handle=None,
            upload=True, window_slice=None, host_overlap=None, group_done=None, groups=None, units_done=None,
            want_uncert=False):
    """Run the hot path for a list of bands on one GPU.

    window_slice=(k, n): process only the k-th of n contiguous window slices of every band (window
    sharding across GPUs); rows outside the slice stay zero, ``nwin``/``t`` describe the whole band.

    data (N, npts) raw traces — a 2-D array or a list of N rows (uploaded from where they lie);
    band_edges [(fmin, fmax), ...]; winlens [seconds per band].
    prefiltered=True: ``data`` is already filtered/tapered (``ltsva`` entry), one band.
    want_uncert=True: also ``res.vel_uncert`` / ``res.baz_uncert`` (nbands, vector_len), the confidence intervals of
    the slowness estimate, computed on the GPU behind each unit's solve (``nbls_set_uncertainty``).

    Host/GPU overlap: the bands are cut into ``groups`` contiguous groups (``pipeline_groups``), each an
    asynchronous pass on its own handle of the same GPU, queued as soon as its filters are designed.
    ``host_overlap(res)`` runs once everything is queued (filter responses, key strings: host work that needs
    no GPU result); ``group_done(res, b0, b1)`` runs as soon as the rows of bands [b0, b1) have landed, while
    later groups are still being computed (the caller builds its dictionary there).
    More bands than fit in HBM at once are processed in consecutive rounds.

    Round 4, where it pays (``stream_pays``: LTS calls of 16 000 units and more; ``groups`` given or ``NBLS_PIPELINE_GROUPS`` > 1
    select the band groups above, smaller and OLS calls are one pass fetched in one piece): ONE pass on one handle whose unit batches — consecutive (band, window) units, each a complete
    correlate -> solve -> pack chain on the GPU — copy their rows into a pinned host mirror as they finish
    (``nbls_stream_results``).  ``units_done(res, u0, u1)`` runs as soon as the rows of the units [u0, u1) (flat index
    over all bands of the call, band-major) are in ``res``, while the GPU works on the next batch; without it
    ``group_done`` is called for the bands a batch completes."""
    nchans, npts = _shape_of(data)
    nb = len(band_edges)
    cap = max_bands_per_pass(nchans, npts)
    # the trace goes up (a blocking copy from pageable memory, inside the library: the GIL is released) on a helper
    # thread while this thread designs the first group's filters AND plans its pass: the handle knows the trace's
    # shape (set_trace_shape), only nbls_execute needs the samples
    uploader = None
    upload_error = []
    resident = False
    row_pipeline = False
    if upload and handle is None and (cap >= 1 or prefiltered):
        rk = getattr(get_handle(device, 0), 'resident_key', None)           # engine.resident_trace: these very buffers are in HBM already
        resident = rk is not None and rk == _trace_key(data, fs)
    if upload and handle is None and (cap >= 1 or prefiltered) and UPLOAD_OVERLAP and not resident:
        h0 = get_handle(device, 0)
        up_rows = list(np.ascontiguousarray(data, dtype=np.float64)) if isinstance(data, np.ndarray) else data
        h0.set_trace_shape(nchans, npts, fs)
        row_pipeline = row_pipeline_for(nchans, npts) and hasattr(h0, 'expect_upload')
        if row_pipeline:
            h0.expect_upload()                    # the pass will be queued while the rows are still going up

        def _upload():
            try:
                h0.upload_rows(up_rows)
            except BaseException as e:            # re-raised on the calling thread below
                upload_error.append(e)
        uploader = _upload_worker().submit(_upload)      # (a thread kept between calls: starting one costs 0.1 ms of the 1.3 ms the copy takes)

    # (started before anything else of the call: from here on every way out joins it)
    try:
        W, inc, nwin = [np.empty(nb, dtype=t) for t in (np.int32, np.int32, np.int64)]
        plans = {}                                   # (bands usually share a few window lengths: one plan per length)
        for b in range(nb):
            wl = float(winlens[b])
            if wl not in plans:
                plans[wl] = planner.window_plan(npts, fs, winlens[b], winover)
            W[b], inc[b], nwin[b] = plans[wl]
        if vector_len is None:
            vector_len = max(1, int(nwin.max()))
        if nwin.max() > vector_len:
            raise ValueError('could not broadcast %d windows into result rows of length %d '
                             '(vector_len too small for this band)' % (int(nwin.max()), vector_len))
        check_elements(nchans, alpha)
        if cap < 1 and not prefiltered:            # not even one band's filtered trace fits the HBM budget
            return process_segmented(data, fs, t0_datenum, rij, band_edges, winlens, winover, alpha, filter_type, filter_order,
                                     filter_ripple, vector_len, device, xcorr_impl, want_lag, want_cmax, want_z, host_overlap,
                                     group_done)
        cap = max(1, cap)
        streamed = groups is None and window_slice is None and stream_pays(alpha, nwin, nchans * (nchans - 1) // 2)
        one_pass = (prefiltered or handle is not None or not upload or (alpha >= 1.0 and not os.environ.get('NBLS_PIPELINE_GROUPS'))) and groups is None
        ngroups = 1 if one_pass or streamed else (groups or pipeline_groups(nwin, nchans * (nchans - 1) // 2))     # (OLS: nothing for the host to do per group)
        ngroups = max(1, min(ngroups, nb))
        # contiguous band groups by unit count.  engine.PIPELINE_SPLIT = (0.15, 0.5, 0.35): explicit shares (a small first
        # group gets the GPU started sooner, a small last group leaves less dictionary work after the GPU has finished)
        split = PIPELINE_SPLIT
        if split and groups is None and ngroups > 1:
            shares = [max(0.0, float(x)) for x in split]
            ngroups = max(1, min(len(shares), nb))
            shares = np.cumsum(shares[:ngroups]) / max(1e-30, float(np.sum(shares[:ngroups])))
        else:
            # mildly decreasing shares (0.41 / 0.33 / 0.26 for three groups): what is left to do on the host after the
            # GPU has finished is the dictionary of the LAST group (measured: 23.1 -> 22.6 ms per cfg-3 call)
            wts = 1.0 + 0.3 * np.arange(ngroups - 1, -1, -1)
            shares = np.cumsum(wts) / float(np.sum(wts))
        cum = np.concatenate(([0], np.cumsum(nwin)))
        cuts = [0]
        for g in range(1, ngroups):
            b = int(np.searchsorted(cum, cum[-1] * shares[g - 1]))
            cuts.append(min(max(b, cuts[-1] + 1), nb - (ngroups - g)))
        cuts.append(nb)
        bounds = [(cuts[g], cuts[g + 1]) for g in range(ngroups)]
        if max(b1 - b0 for b0, b1 in bounds) * ngroups > cap:           # HBM budget: consecutive rounds of <= cap bands
            bounds = [(b0, min(b0 + cap, nb)) for b0 in range(0, nb, cap)]
            sequential = True
        else:
            sequential = False

        def upload_done():
            nonlocal uploader
            if uploader is not None:
                uploader.join()
                uploader = None
                if upload_error:
                    raise upload_error[0]
        P = nchans * (nchans - 1) // 2
        MB = (P + 7) // 8
        grids = np.zeros((4, nb, vector_len))
        mask = np.zeros((nb, vector_len, MB), dtype=np.uint8)
        lag = np.zeros((nb, vector_len, P), dtype=np.int32) if want_lag else None
        cmax = np.zeros((nb, vector_len, P)) if want_cmax else None
        z = np.zeros((nb, vector_len, 2)) if want_z else None
        unc = np.zeros((2, nb, vector_len)) if want_uncert else None
        res = BandBatch(vel=grids[0], baz=grids[1], mdccm=grids[2], sigma_tau=grids[3], nwin=nwin.astype(int), t=None,
                        mask=mask, lag=lag, cmax=cmax, z=z, sos=[], W=W, inc=inc, pair_idx=None, xij=None,
                        nchans=nchans, alpha=alpha, handle=None, lts=alpha < 1.0, fs=fs,
                        vel_uncert=None if unc is None else unc[0], baz_uncert=None if unc is None else unc[1])

        deferred = []                                 # rounds collected before host_overlap has run (sequential rounds)

        cum_units = np.concatenate(([0], np.cumsum(nwin))).astype(np.int64)
        done_band = [0]                               # bands [0, done_band) have been reported through group_done

        def collect_streamed(h, b0, b1, notify):
            # the rows arrive batch by batch (pinned mirror of the result block): copy each batch's cells out and tell
            # the caller, while the GPU is busy with the next batch
            for k in range(h.result_batches()):
                u0, u1, c0, c1, gsrc, msrc = h.wait_result_batch(k)
                if c1 > c0:
                    for g in range(4):
                        grids[g, b0:b1].reshape(-1)[c0:c1] = gsrc[g, c0:c1]
                    mask[b0:b1].reshape(-1, MB)[c0:c1] = msrc[c0:c1]
                if not notify:
                    continue
                g0, g1 = int(cum_units[b0]) + u0, int(cum_units[b0]) + u1
                if units_done is not None:
                    if g1 > g0:
                        units_done(res, g0, g1)
                elif group_done is not None:
                    bd = int(np.searchsorted(cum_units, g1, side='right')) - 1     # bands complete up to unit g1
                    bd = min(max(bd, done_band[0]), b1)
                    if g1 >= cum_units[b1]:
                        bd = b1
                    if bd > done_band[0]:
                        group_done(res, done_band[0], bd)
                        done_band[0] = bd
            if getattr(h, 'profiling', False):
                h.sync()                               # (turns the pass's events into ``timings()``)

        def collect(h, b0, b1, notify=True):
            if streamed:
                collect_streamed(h, b0, b1, notify)
            else:
                out = h.fetch_packed()                    # waits for that pass; ONE D2H copy (grids + weight mask)
                grids[:, b0:b1] = np.stack((out['vel'], out['baz'], out['mdccm'], out['sigma_tau']))
                mask[b0:b1] = out['mask']
            if want_lag or want_cmax or want_z:
                ext = h.fetch(want_lag=want_lag, want_cmax=want_cmax, want_z=want_z, grids=False)
                for name, arr in (('lag', lag), ('cmax', cmax), ('z', z)):
                    if arr is not None:
                        arr[b0:b1] = ext[name]
            if want_uncert:
                unc[0, b0:b1], unc[1, b0:b1] = h.fetch_uncertainty()
            if not notify:
                deferred.append((b0, b1))
            elif streamed:
                pass                                   # (told batch by batch above)
            elif units_done is not None:
                if cum_units[b1] > cum_units[b0]:
                    units_done(res, int(cum_units[b0]), int(cum_units[b1]))
            elif group_done is not None:
                group_done(res, b0, b1)

        def finish_skeleton(prep):
            res.pair_idx, res.xij = prep.pair_idx, prep.xij
            tt = np.zeros((nb, vector_len))
            rows_t = {}                              # (one row of window times per distinct window plan)
            for b in range(nb):
                key = (int(W[b]), int(inc[b]), int(nwin[b]))
                if key not in rows_t:
                    rows_t[key] = window_times(t0_datenum, fs, *key)
                tt[b, :nwin[b]] = rows_t[key]
            res.t = tt

    except BaseException:
        if uploader is not None:
            uploader.join()
        raise
    launched = []
    prep = None
    try:
        for g, (b0, b1) in enumerate(bounds):
            prep = prepare(nchans, npts, fs, rij, band_edges[b0:b1], winlens[b0:b1], winover, alpha, filter_type,
                           filter_order, filter_ripple, vector_len, prefiltered, common=prep)
            res.sos.extend(prep.sos_ret)
            h = handle if handle is not None else get_handle(device, 0 if sequential else g)
            if sequential and launched:           # one handle, one plan at a time: finish the previous round first
                collect(*launched.pop(), notify=False)     # (group_done waits for host_overlap: replayed below)
            early = (uploader is not None and g == 0) or (resident and (g == 0 or sequential))
            # the groups finish in the order they were queued (GPU-side ordering of their correlation stages): the
            # dictionary of group k is built while groups k+1.. are still running.  Left to itself the GPU shares
            # itself between the passes and all of them land together at the end (stream priorities alone did the
            # job on some boxes and not on others)
            ordered = launched and not sequential and GROUP_ORDER
            joins = early and not resident               # this launch rides on the upload thread's rows
            try:
                launch(h, data, prep, upload=upload, window_slice=window_slice, uncert=want_uncert,
                       xcorr_impl=xcorr_impl, trace_from=launched[0][0] if (launched and not sequential) else None,
                       trace_ready=early, after=launched[-1][0] if ordered else None,
                       before_execute=upload_done if (joins and not row_pipeline) else None, stream=streamed)
            except BaseException:
                if joins and uploader is not None:       # the pass could not be queued: the upload's own error is the cause, if it has one
                    uploader.join()
                    uploader = None
                    if upload_error:
                        raise upload_error[0]
                raise
            if joins:
                upload_done()                            # (row_pipeline: the pass is queued, the library filters the rows as they land)
            launched.append((h, b0, b1))
            res.handle = h
    finally:
        if uploader is not None:                  # prepare() / plan raised: do not leave the copy running behind the caller
            uploader.join()
    # everything is queued: host work that needs no GPU result hides behind the passes
    release_deferred()
    finish_skeleton(prep)
    if host_overlap is not None:
        host_overlap(res)
    for b0, b1 in deferred:                       # rounds that landed before the skeleton existed, in band order
        if units_done is not None:
            if cum_units[b1] > cum_units[b0]:
                units_done(res, int(cum_units[b0]), int(cum_units[b1]))
        elif group_done is not None:
            group_done(res, b0, b1)
            done_band[0] = b1
    for item in launched:
        collect(*item)
    return res


def time_keys(t, nwin, prefixes=None):
    """The ``stdict`` key text of every (band, window): prefix + ``str(numpy.float64 time)`` — repr of a
    Python float is the same shortest round-trip text.  -> ONE flat list of strings, bands in order,
    ``nwin[b]`` keys per band.  Needs no GPU result, so the band loop computes it while the pass is running."""
    if _hostext is not None:
        return _hostext.time_keys(np.ascontiguousarray(t, dtype=np.float64), np.ascontiguousarray(nwin, dtype=np.int64),
                                  None if prefixes is None else list(prefixes))
    return _py_time_keys(t, nwin, prefixes)


def time_key_text(t, nwin, prefixes=None):
    """``time_keys`` without the string objects: the key TEXT of every (band, window) as a ``(text (K, 40) uint8,
    length (K,) uint8)`` pair, formatted in C++ with the GIL released on a few threads.  ``stdict_from_mask`` takes
    the pair in place of the key list and makes a string only for the windows that get an entry.  Without the
    helper module this is ``time_keys`` (a list)."""
    if _hostext is not None and hasattr(_hostext, 'time_key_text'):
        return _hostext.time_key_text(np.ascontiguousarray(t, dtype=np.float64), np.ascontiguousarray(nwin, dtype=np.int64),
                                      None if prefixes is None else list(prefixes), _key_threads())
    return _py_time_keys(t, nwin, prefixes)


def _key_threads():
    if KEY_THREADS:
        return max(1, int(KEY_THREADS))
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(4, n - 1))


def n_keys(keys):
    """Number of keys in a ``time_keys`` list or a ``time_key_text`` pair."""
    return len(keys[1]) if isinstance(keys, tuple) else len(keys)


def _py_time_keys(t, nwin, prefixes=None):
    out = []
    for b in range(len(nwin)):
        p = '' if prefixes is None else prefixes[b]
        reps = map(repr, np.asarray(t[b, :int(nwin[b])], dtype=np.float64).tolist())
        out.extend([p + s for s in reps] if p else reps)
    return out


def new_stdict(nkeys):
    """An empty dictionary with room for ``nkeys`` entries (no re-hashing while a call's ~5*10^4 keys go in)."""
    if _hostext is not None:
        return _hostext.new_dict(int(nkeys) + 1)
    return {}


def new_pattern_cache():
    """Holder of the shared value arrays for the ``stdict_from_mask`` calls of ONE pipelined call (one per band group):
    a dropped-pair pattern met in an earlier group is not built again.  None without the helper module."""
    if _hostext is not None and hasattr(_hostext, 'new_pattern_cache'):
        return _hostext.new_pattern_cache()
    return None


def stdict_from_mask(mask, nwin, pair_idx, nchans, keys, into=None, k0=0, cache=None, units=None):
    """lts_array's dropped-element dictionary for ALL bands of a pass from the packed weight mask
    (B, VL, ceil(P/8)): key (``keys[b][w]``, see ``time_keys``) -> 1-based element numbers of both
    members of every zero-weight pair (first members, then second members), only for windows that
    dropped something; ``'size'`` -> number of elements.  Entries are inserted in (band, window) order as
    the reference's loops do (narrow_band_least_squares.py:114-124; ``'size'`` therefore sits right after
    the first band's entries).

    Windows that dropped the SAME set of pairs share ONE read-only array (the reference makes a fresh
    array per window; the values are equal, and an in-place write raises instead of aliasing): creating
    ~5*10^4 tiny arrays per call costs as much host time as the whole GPU pass.  With the C++ helper
    module built, one C++ pass does the work; the NumPy form below is its equivalent.

    ``into`` / ``k0``: update an existing dictionary with the bands of one group of a pipelined call (``k0`` =
    index in ``keys`` of this mask's first window); 'size' is only placed by the call that starts the dictionary.

    ``units=(u0, u1)``: enter only the units [u0, u1) of the flattened (band, window) order of ``mask`` / ``nwin`` (the
    batches of a streamed pass; consecutive ranges in ascending order); 'size' goes in with the range that holds the
    first band's last window."""
    if _hostext is not None:
        u0, u1 = (-1, -1) if units is None else (int(units[0]), int(units[1]))
        return _hostext.build_stdict(np.ascontiguousarray(mask, dtype=np.uint8), np.ascontiguousarray(nwin, dtype=np.int64),
                                     np.ascontiguousarray(pair_idx, dtype=np.int32), int(nchans), keys, into, int(k0), cache, u0, u1)
    return _py_stdict_from_mask(mask, nwin, pair_idx, nchans, keys, into, k0, units)


def _py_stdict_from_mask(mask, nwin, pair_idx, nchans, keys, into=None, k0=0, units=None):
    if units is not None:
        # a unit range: the bands it touches, cut at the end of the first band ('size' follows that band's entries)
        u0, u1 = int(units[0]), int(units[1])
        nw = np.asarray(nwin, dtype=np.int64)
        cum = np.concatenate(([0], np.cumsum(nw)))
        stdict = {} if into is None else into
        B, VL, MB = mask.shape
        flat_keys = keys
        if isinstance(keys, tuple):
            text, length = keys
            flat_keys = [bytes(text[i, :length[i]]).decode('ascii') for i in range(len(length))]
        full = np.packbits(np.ones(len(pair_idx), dtype=np.uint8), bitorder='little')
        pidx = np.asarray(pair_idx)
        for b in range(B):
            lo, hi = max(u0, int(cum[b])), min(u1, int(cum[b + 1]))
            for u in range(lo, hi):
                m = mask[b, u - int(cum[b])]
                if ((m & full) != full).any():
                    cols = np.nonzero(np.unpackbits(m, bitorder='little')[:len(pidx)] == 0)[0]
                    arr = np.concatenate((pidx[cols, 0] + 1, pidx[cols, 1] + 1)).astype((pidx[:1, 0] + 1).dtype, copy=False)
                    arr.flags.writeable = False
                    stdict[flat_keys[k0 + u]] = arr
            if b == 0 and ((u0 < nw[0] <= u1) or (nw[0] == 0 and u0 == 0)):
                stdict['size'] = nchans
        return stdict
    if isinstance(keys, tuple):                                  # (text, length) of time_key_text
        text, length = keys
        keys = [bytes(text[i, :length[i]]).decode('ascii') for i in range(len(length))]
    pair_idx = np.asarray(pair_idx)
    P = len(pair_idx)
    B, VL, MB = mask.shape
    nwin = np.asarray(nwin, dtype=np.int64)
    valid = np.arange(VL)[None, :] < nwin[:, None]
    m = mask[valid]                                              # (U, MB), (band, window) order
    full = np.packbits(np.ones(P, dtype=np.uint8), bitorder='little')
    hit = ((m & full[None, :]) != full[None, :]).any(axis=1)
    stdict = {} if into is None else into
    fresh = len(stdict) == 0
    keys = keys[k0:k0 + len(hit)] if (k0 or len(keys) != len(hit)) else keys
    n = int(np.count_nonzero(hit))
    if n:
        sel = np.ascontiguousarray(m[hit])
        if MB <= 8:                                              # pattern code of every window
            code = np.zeros(n, dtype=np.uint64)
            for i in range(MB):
                code |= sel[:, i].astype(np.uint64) << np.uint64(8 * i)
        else:
            code = sel.view(np.dtype((np.void, MB))).ravel()
        _, first, inv = np.unique(code, return_index=True, return_inverse=True)
        one = (pair_idx[:1, 0] + 1).dtype
        pats = []
        for row in np.unpackbits(sel[first], axis=1, bitorder='little')[:, :P]:
            cols = np.nonzero(row == 0)[0]
            arr = np.concatenate((pair_idx[cols, 0] + 1, pair_idx[cols, 1] + 1)).astype(one, copy=False)
            arr.flags.writeable = False
            pats.append(arr)
        pieces = operator.itemgetter(*inv.ravel().tolist())(pats) if n > 1 else (pats[0],)
        items = zip(itertools.compress(keys, hit.tolist()), pieces)
        # key order of the reference's merge loop: the first band's time keys, 'size', then the other bands
        if fresh:
            stdict.update(itertools.islice(items, int(np.count_nonzero(hit[:int(nwin[0])]))))
            stdict['size'] = nchans
        stdict.update(items)
    if fresh:
        stdict['size'] = nchans
    return stdict


def stdict_from_weights(weights_row, nwin, t_row, pair_idx, nchans, prefix=''):
    """Single-band form of ``stdict_from_mask`` taking unpacked weights (nwin.., P) uint8 and the window
    times: key ``prefix + str(t)`` (``prefix`` = the band prefix of narrow_band_least_squares.py:114-124)."""
    nwin = int(nwin)
    w = np.asarray(weights_row[:nwin]) != 0
    mask = np.packbits(w, axis=-1, bitorder='little')[None, :, :]
    keys = time_keys(np.asarray(t_row, dtype=np.float64)[None, :nwin], [nwin], [prefix] if prefix else None)
    return stdict_from_mask(mask, [nwin], pair_idx, nchans, keys)
