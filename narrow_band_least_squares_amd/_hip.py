"""ctypes binding of ``libnbls_hip.so`` (C ABI: ``include/nbls.h``).

There is no CPU fallback: if the library is missing or no HIP device can be opened, the
functions of this package raise.  No PyTorch is involved on this path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NBLS_LIB: another build of the library (tools/ use the developer build, csrc/libnbls_hip_dev.so)
LIB_PATH = os.environ.get('NBLS_LIB') or os.path.join(_HERE, 'csrc', 'libnbls_hip.so')

EXPORTS = [
    'nbls_version', 'nbls_device_count', 'nbls_create', 'nbls_destroy', 'nbls_last_error', 'nbls_set_trace',
    'nbls_set_geometry', 'nbls_plan', 'nbls_execute', 'nbls_execute_stages', 'nbls_execute_after', 'nbls_set_trace_shape', 'nbls_upload_rows', 'nbls_sync',
    'nbls_fetch', 'nbls_fetch_filtered', 'nbls_device_results', 'nbls_set_profiling',
    'nbls_get_timings', 'nbls_run', 'nbls_probe_mfma_f64', 'nbls_probe_mfma_i8', 'nbls_debug_screen_stats', 'nbls_set_window_ranges', 'nbls_debug_screen_stamps', 'nbls_debug_lts_stamps',
    'nbls_set_trace_rows', 'nbls_result_layout', 'nbls_fetch_packed', 'nbls_comm_init_all', 'nbls_comm_unique_id',
    'nbls_comm_init_rank', 'nbls_reserve_results', 'nbls_comm_gather', 'nbls_comm_destroy', 'nbls_set_option',
    'nbls_developer_build', 'nbls_set_trace_from', 'nbls_debug_lts_coop_breakdown', 'nbls_filter_segment',
    'nbls_set_filtered', 'nbls_load_result_block', 'nbls_stream_results', 'nbls_result_batches', 'nbls_wait_result_batch',
    'nbls_comm_set_library', 'nbls_set_uncertainty', 'nbls_fetch_uncertainty', 'nbls_expect_upload', 'nbls_abort_upload',
]

NBLS_ERR_ARG, NBLS_ERR_STATE, NBLS_ERR_GEOMETRY = -1, -2, -3
NBLS_ERR_HIP, NBLS_ERR_NOMEM, NBLS_ERR_UNSUPPORTED, NBLS_ERR_COMM = -4, -5, -6, -7


class NblsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libnbls_hip error %d: %s' % (code, msg))
        self.code = code


class LtsParams(C.Structure):
    _fields_ = [
        ('alpha', C.c_double), ('h', C.c_int32), ('nstarts', C.c_int32),
        ('starts', C.POINTER(C.c_int32)), ('csteps', C.c_int32), ('csteps2', C.c_int32),
        ('ncand', C.c_int32), ('xij_mad', C.c_double * 2), ('raw_factor', C.c_double),
        ('rew_table', C.POINTER(C.c_double)), ('quantile', C.c_double), ('zero_scale', C.c_double),
    ]


class Timings(C.Structure):
    _fields_ = [('filter_ms', C.c_double), ('xcorr_ms', C.c_double), ('solve_ms', C.c_double),
                ('total_ms', C.c_double), ('xcorr_launches', C.c_int64), ('quantize_ms', C.c_double),
                ('screen_ms', C.c_double), ('verify_ms', C.c_double), ('xcorr_impl', C.c_int32),
                ('xcorr_fallback_bands', C.c_int32)]


_lib = None


def load_library(path=None):
    """Load (once) and return the ctypes library; raises ImportError if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError('%s not found: build it with `make -C %s` (or __graft_entry__.build()); '
                          'this package has no CPU fallback' % (p, os.path.dirname(p)))
    lib = C.CDLL(p)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    u8p = C.POINTER(C.c_uint8)
    vp = C.c_void_p
    lib.nbls_version.restype = C.c_int
    lib.nbls_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.nbls_destroy.argtypes = [vp]
    lib.nbls_destroy.restype = None
    lib.nbls_last_error.argtypes = [vp]
    lib.nbls_last_error.restype = C.c_char_p
    lib.nbls_set_trace.argtypes = [vp, dp, C.c_int32, C.c_int64, C.c_double]
    lib.nbls_set_trace_rows.argtypes = [vp, C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_double]
    lib.nbls_result_layout.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.nbls_fetch_packed.argtypes = [vp, C.c_void_p, C.c_int64]
    lib.nbls_comm_init_all.argtypes = [C.POINTER(vp), C.c_int32]
    lib.nbls_comm_unique_id.argtypes = [C.c_void_p, C.c_int32]
    lib.nbls_comm_init_rank.argtypes = [vp, C.c_void_p, C.c_int32, C.c_int32]
    lib.nbls_reserve_results.argtypes = [vp, C.c_int64]
    lib.nbls_load_result_block.argtypes = [vp, C.c_void_p, C.c_int64]
    lib.nbls_comm_gather.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int64]
    lib.nbls_comm_destroy.argtypes = [vp]
    lib.nbls_expect_upload.argtypes = [vp]
    lib.nbls_abort_upload.argtypes = [vp]
    lib.nbls_comm_set_library.argtypes = [C.c_char_p, C.c_int32]
    lib.nbls_set_uncertainty.argtypes = [vp, dp]
    lib.nbls_fetch_uncertainty.argtypes = [vp, dp, dp]
    lib.nbls_stream_results.argtypes = [vp, C.c_int32]
    lib.nbls_result_batches.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.nbls_wait_result_batch.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_void_p)]
    lib.nbls_set_trace_from.argtypes = [vp, vp]
    lib.nbls_filter_segment.argtypes = [vp, C.c_int32, dp, dp]
    lib.nbls_set_filtered.argtypes = [vp, C.c_int32, dp]
    lib.nbls_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.nbls_set_geometry.argtypes = [vp, dp, ip, dp, C.c_int32]
    plan_args = [vp, C.c_int32, dp, C.c_int32, C.c_int32, dp, dp, C.c_int32, ip, ip, C.c_int32,
                 C.POINTER(LtsParams), C.c_int32]
    lib.nbls_plan.argtypes = plan_args
    lib.nbls_execute.argtypes = [vp]
    lib.nbls_execute_stages.argtypes = [vp, C.c_int32]
    lib.nbls_execute_after.argtypes = [vp, vp]
    lib.nbls_set_trace_shape.argtypes = [vp, C.c_int32, C.c_int64, C.c_double]
    lib.nbls_upload_rows.argtypes = [vp, C.POINTER(C.c_void_p), C.c_int32, C.c_int64]
    lib.nbls_sync.argtypes = [vp]
    fetch_args = [dp, dp, dp, dp, ip, ip, dp, u8p, dp]
    lib.nbls_fetch.argtypes = [vp] + fetch_args
    lib.nbls_fetch_filtered.argtypes = [vp, C.c_int32, dp]
    lib.nbls_device_results.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.nbls_set_profiling.argtypes = [vp, C.c_int32]
    lib.nbls_get_timings.argtypes = [vp, C.POINTER(Timings)]
    lib.nbls_run.argtypes = plan_args + fetch_args
    lib.nbls_probe_mfma_f64.argtypes = [vp, dp, dp, dp]
    lib.nbls_probe_mfma_i8.argtypes = [vp, ip, ip, ip]
    lib.nbls_debug_screen_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.nbls_set_window_ranges.argtypes = [vp, C.c_int32, ip, ip]
    lib.nbls_debug_screen_stamps.argtypes = [vp, dp]
    lib.nbls_debug_lts_stamps.argtypes = [vp, dp]
    lib.nbls_debug_lts_coop_breakdown.argtypes = [vp, dp]
    for name in EXPORTS:
        if name not in ('nbls_destroy', 'nbls_last_error'):
            getattr(lib, name).restype = C.c_int
    if path is None:
        _lib = lib
    return lib


def _dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _u8ptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Handle:
    """One GPU, one stream, one host thread at a time."""

    def __init__(self, device_id=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.nbls_create(int(device_id), C.byref(h))
        if rc != 0:
            raise NblsError(rc, self.lib.nbls_last_error(None).decode())
        self._h = h
        self.device_id = int(device_id)
        self.nchans = self.npts = self.npairs = 0
        self.nbands = self.vector_len = 0
        self._keep = []
        self.profiling = False
        self.resident_key = None        # engine.resident_trace: what the trace in HBM was uploaded from (any new trace clears it)

    def close(self):
        if getattr(self, '_h', None):
            self.lib.nbls_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            msg = self.lib.nbls_last_error(self._h).decode()
            if rc == NBLS_ERR_GEOMETRY:
                raise RuntimeError(msg)
            if rc in (NBLS_ERR_ARG, NBLS_ERR_UNSUPPORTED):
                raise ValueError(msg)
            raise NblsError(rc, msg)

    def set_trace(self, data, fs):
        self.resident_key = None
        data = _f64(data)
        if data.ndim != 2:
            raise ValueError('trace must be (nchans, npts)')
        self._chk(self.lib.nbls_set_trace(self._h, _dptr(data), data.shape[0], data.shape[1], float(fs)))
        self.nchans, self.npts, self.fs = data.shape[0], data.shape[1], float(fs)

    def set_trace_rows(self, rows, fs):
        """rows: one 1-D float64 C-contiguous array per channel, all of the same length (uploaded
        straight from where they are: no packed host copy)."""
        self.resident_key = None
        npts = len(rows[0])
        keep = []
        for r in rows:
            r = np.asarray(r)
            if r.dtype != np.float64 or not r.flags.c_contiguous:
                r = np.ascontiguousarray(r, dtype=np.float64)
            if r.ndim != 1 or len(r) != npts:
                raise ValueError('All traces must have the same number of samples.')
            keep.append(r)
        ptrs = (C.c_void_p * len(keep))(*[r.ctypes.data for r in keep])
        self._chk(self.lib.nbls_set_trace_rows(self._h, ptrs, len(keep), npts, float(fs)))
        self.nchans, self.npts, self.fs = len(keep), npts, float(fs)

    def set_trace_shape(self, nchans, npts, fs):
        """Declare the trace (no samples yet): geometry and plan may follow while ``upload_rows`` runs on another thread."""
        self.resident_key = None
        self._chk(self.lib.nbls_set_trace_shape(self._h, int(nchans), int(npts), float(fs)))
        self.nchans, self.npts, self.fs = int(nchans), int(npts), float(fs)

    def expect_upload(self):
        """Announce (on the thread that goes on to plan and execute) that ``upload_rows`` is about to run on another
        thread: ``execute`` may then be called while the rows are still going up, the pass filters them as they land."""
        self._chk(self.lib.nbls_expect_upload(self._h))

    def upload_rows(self, rows):
        """The samples of the declared trace: one 1-D float64 C-contiguous array per channel."""
        self.resident_key = None
        try:
            keep = []
            for r in rows:
                r = np.asarray(r)
                if r.dtype != np.float64 or not r.flags.c_contiguous:
                    r = np.ascontiguousarray(r, dtype=np.float64)
                if r.ndim != 1 or len(r) != self.npts:
                    raise ValueError('All traces must have the same number of samples.')
                keep.append(r)
            if len(keep) != self.nchans:
                raise ValueError('upload_rows: %d rows for a trace declared with %d channels' % (len(keep), self.nchans))
            ptrs = (C.c_void_p * len(keep))(*[r.ctypes.data for r in keep])
        except BaseException:
            self.lib.nbls_abort_upload(self._h)        # (a pass that was queued on the announced rows fails instead of waiting)
            raise
        self._chk(self.lib.nbls_upload_rows(self._h, ptrs, len(keep), self.npts))

    def set_trace_from(self, other):
        """The trace of another handle on the same GPU, copied device-to-device."""
        self.resident_key = None
        self._chk(self.lib.nbls_set_trace_from(self._h, other._h))
        self.nchans, self.npts, self.fs = other.nchans, other.npts, other.fs

    def set_geometry(self, xij, pair_idx, xpinv):
        xij = _f64(xij)
        pair_idx = np.ascontiguousarray(pair_idx, dtype=np.int32)
        xpinv = _f64(xpinv)
        self._chk(self.lib.nbls_set_geometry(self._h, _dptr(xij), _iptr(pair_idx), _dptr(xpinv), xij.shape[0]))
        self.npairs = xij.shape[0]

    def plan(self, sos, zero_phase, taper_left, taper_right, winlen, wininc, vector_len, lts=None,
             xcorr_impl=0):
        """sos: (B, S, 6) or None (unfiltered single band).  lts: dict from planner.lts_plan()."""
        winlen = np.ascontiguousarray(winlen, dtype=np.int32)
        wininc = np.ascontiguousarray(wininc, dtype=np.int32)
        nb = len(winlen)
        if sos is None:
            sos_p, nsec = None, 0
        else:
            sos = _f64(sos)
            if sos.ndim != 3 or sos.shape[0] != nb or sos.shape[2] != 6:
                raise ValueError('sos must be (nbands, nsections, 6)')
            sos_p, nsec = _dptr(sos), sos.shape[1]
        tl = _f64(taper_left if taper_left is not None else np.zeros(0))
        tr = _f64(taper_right if taper_right is not None else np.zeros(0))
        if len(tl) != len(tr):
            raise ValueError('taper ramps must have equal length')
        lp = None
        keep = [sos, tl, tr, winlen, wininc]
        if lts is not None:
            starts = np.ascontiguousarray(lts['starts'], dtype=np.int32)
            rew = _f64(lts['rew_table'])
            p = LtsParams()
            p.alpha = float(lts['alpha'])
            p.h = int(lts['h'])
            p.nstarts = int(starts.shape[0])
            p.starts = _iptr(starts)
            p.csteps = int(lts['csteps'])
            p.csteps2 = int(lts['csteps2'])
            p.ncand = int(lts['ncand'])
            p.xij_mad[0] = float(lts['xij_mad'][0])
            p.xij_mad[1] = float(lts['xij_mad'][1])
            p.raw_factor = float(lts['raw_factor'])
            p.rew_table = _dptr(rew)
            p.quantile = float(lts['quantile'])
            p.zero_scale = float(lts['zero_scale'])
            lp = C.byref(p)
            keep += [starts, rew, p]
        self._chk(self.lib.nbls_plan(self._h, nb, sos_p, nsec, int(bool(zero_phase)), _dptr(tl), _dptr(tr),
                                     len(tl), _iptr(winlen), _iptr(wininc), int(vector_len), lp,
                                     int(xcorr_impl)))
        self.nbands, self.vector_len = nb, int(vector_len)
        self._nsec = nsec

    def set_option(self, key, value):
        """Per-handle implementation switch (``nbls_set_option``; identical results unless the library is the
        developer build and the key is one of its timing switches)."""
        self._chk(self.lib.nbls_set_option(self._h, key.encode(), int(value)))

    def reserve_results(self, nbytes):
        """Minimum allocation of the result block for the next plans (equal-sized gather blocks)."""
        self._chk(self.lib.nbls_reserve_results(self._h, int(nbytes)))

    def load_result_block(self, block):
        """Put a result block assembled on the host (several HBM rounds of one rank's share) back where the gather
        sends from.  ``block``: contiguous uint8 array in the layout of ``nbls_result_layout`` for all of the rank's bands."""
        block = np.ascontiguousarray(block, dtype=np.uint8)
        self._chk(self.lib.nbls_load_result_block(self._h, block.ctypes.data, block.nbytes))

    def set_window_ranges(self, first=None, count=None):
        """Per-band window slices for the next plan(s); None resets to "all windows"."""
        if first is None:
            self._chk(self.lib.nbls_set_window_ranges(self._h, 0, None, None))
            return
        first = np.ascontiguousarray(first, dtype=np.int32)
        count = np.ascontiguousarray(count, dtype=np.int32)
        self._chk(self.lib.nbls_set_window_ranges(self._h, len(first), _iptr(first), _iptr(count)))

    def execute(self, stages=7, after=None):
        """Queue the pass.  ``after``: another handle of the same GPU whose pass was queued before — this pass's
        correlation stage then starts when that one's is through (``nbls_execute_after``)."""
        if after is not None:
            self._chk(self.lib.nbls_execute_after(self._h, after._h))
        else:
            self._chk(self.lib.nbls_execute_stages(self._h, int(stages)))

    def sync(self):
        self._chk(self.lib.nbls_sync(self._h))

    def fetch(self, want_lag=False, want_cmax=False, want_weights=False, want_z=False, grids=True):
        B, VL, P = self.nbands, self.vector_len, self.npairs
        if grids:
            g = np.empty((4, B, VL))               # one block: nbls_fetch moves the four grids in one copy
            out = dict(vel=g[0], baz=g[1], mdccm=g[2], sigma_tau=g[3], nwin=np.empty(B, dtype=np.int32))
        else:
            out = dict(vel=None, baz=None, mdccm=None, sigma_tau=None, nwin=np.empty(B, dtype=np.int32))
        lag = np.empty((B, VL, P), dtype=np.int32) if want_lag else None
        cmax = np.empty((B, VL, P)) if want_cmax else None
        wts = np.empty((B, VL, P), dtype=np.uint8) if want_weights else None
        z = np.empty((B, VL, 2)) if want_z else None
        self._chk(self.lib.nbls_fetch(self._h, _dptr(out['vel']), _dptr(out['baz']), _dptr(out['mdccm']),
                                      _dptr(out['sigma_tau']), _iptr(out['nwin']), _iptr(lag), _dptr(cmax),
                                      _u8ptr(wts), _dptr(z)))
        out.update(lag=lag, cmax=cmax, weights=wts, z=z)
        return out

    def fetch_packed(self):
        """One D2H copy of the result block -> dict(vel, baz, mdccm, sigma_tau (B, VL) float64 views of one
        buffer, mask (B, VL, ceil(P/8)) uint8: bit k & 7 of byte k >> 3 = LTS weight of pair k)."""
        lay = (C.c_int64 * 4)()
        self._chk(self.lib.nbls_result_layout(self._h, lay))
        cells, mb, total, moff = lay[0], lay[1], lay[2], lay[3]
        buf = np.empty(total // 8 + 1, dtype=np.float64)       # 8-byte aligned
        self._chk(self.lib.nbls_fetch_packed(self._h, buf.ctypes.data, total))
        B, VL = self.nbands, self.vector_len
        grids = buf[:4 * cells].reshape(4, B, VL)
        mask = buf.view(np.uint8)[moff:moff + cells * mb].reshape(B, VL, mb)
        return dict(vel=grids[0], baz=grids[1], mdccm=grids[2], sigma_tau=grids[3], mask=mask)

    def set_uncertainty(self, eig6=None):
        """``eig6`` = (lambda_0, lambda_1, R00, R01, R10, R11) of the co-array (``planner.uncertainty_frame``): the
        next plans also compute ltsva's confidence intervals on the GPU; None switches that off."""
        e = None if eig6 is None else _f64(eig6)
        self._chk(self.lib.nbls_set_uncertainty(self._h, _dptr(e)))

    def fetch_uncertainty(self):
        """-> (vel_uncert, baz_uncert), each (nbands, vector_len)."""
        out = np.empty((2, self.nbands, self.vector_len))
        self._chk(self.lib.nbls_fetch_uncertainty(self._h, _dptr(out[0]), _dptr(out[1])))
        return out[0], out[1]

    def stream_results(self, on=True):
        """The next passes deliver their rows batch by batch into a pinned host mirror of the result block
        (``nbls_stream_results``); see ``result_batches`` / ``wait_result_batch``."""
        self._chk(self.lib.nbls_stream_results(self._h, int(bool(on))))

    def result_batches(self):
        n = C.c_int32()
        self._chk(self.lib.nbls_result_batches(self._h, C.byref(n)))
        return n.value

    def wait_result_batch(self, k):
        """Wait for batch ``k`` of the queued pass -> (u0, u1, c0, c1, grids (4, B*VL) float64, mask (B*VL, MB) uint8):
        the batch's units [u0, u1) own the cells [c0, c1) of the two VIEWS of the library's pinned mirror (valid until
        the handle's next pass; cells of batches not yet waited for are undefined)."""
        out = (C.c_int64 * 4)()
        blk = C.c_void_p()
        self._chk(self.lib.nbls_wait_result_batch(self._h, int(k), out, C.byref(blk)))
        cells = self.nbands * self.vector_len
        mb = (self.npairs + 7) // 8
        raw = (C.c_uint8 * (cells * (32 + mb))).from_address(blk.value)
        buf = np.frombuffer(raw, dtype=np.uint8)
        grids = buf[:32 * cells].view(np.float64).reshape(4, cells)
        mask = buf[32 * cells:].reshape(cells, mb)
        return out[0], out[1], out[2], out[3], grids, mask

    def fetch_filtered(self, band):
        out = np.empty((self.nchans, self.npts))
        self._chk(self.lib.nbls_fetch_filtered(self._h, int(band), _dptr(out)))
        return out

    def filter_segment(self, reverse=False, state_in=None, want_state=False):
        """One causal filter pass over the resident segment continuing from ``state_in`` (nbands, nchans, 2*nsections)
        -> the state leaving the segment (or None).  See ``nbls_filter_segment``."""
        si = None if state_in is None else _f64(state_in)
        so = np.empty((self.nbands, self.nchans, 2 * self._nsec)) if want_state else None
        self._chk(self.lib.nbls_filter_segment(self._h, int(bool(reverse)), _dptr(si), _dptr(so)))
        return so

    def set_filtered(self, band, data):
        data = _f64(data)
        if data.shape != (self.nchans, self.npts):
            raise ValueError('filtered band must be (nchans, npts)')
        self._chk(self.lib.nbls_set_filtered(self._h, int(band), _dptr(data)))

    def device_results(self):
        ptrs = (C.c_void_p * 5)()
        nbytes = C.c_int64()
        self._chk(self.lib.nbls_device_results(self._h, ptrs, C.byref(nbytes)))
        return [p or 0 for p in ptrs], nbytes.value

    def set_profiling(self, on=True):
        self._chk(self.lib.nbls_set_profiling(self._h, int(bool(on))))
        self.profiling = bool(on)

    def timings(self):
        t = Timings()
        self._chk(self.lib.nbls_get_timings(self._h, C.byref(t)))
        return dict(filter_ms=t.filter_ms, xcorr_ms=t.xcorr_ms, solve_ms=t.solve_ms,
                    total_ms=t.total_ms, xcorr_launches=t.xcorr_launches, quantize_ms=t.quantize_ms,
                    screen_ms=t.screen_ms, verify_ms=t.verify_ms, xcorr_impl=t.xcorr_impl,
                    xcorr_fallback_bands=t.xcorr_fallback_bands)

    def screen_stamps(self):
        out = np.zeros(10)
        self._chk(self.lib.nbls_debug_screen_stamps(self._h, _dptr(out)))
        return dict(zip(('stage_issue', 'stage_wait', 'compute_wave0', 'wait_other_waves', 'merge_write', 'total',
                         'kloop_cycles_wave0', 'epilogue_cycles_wave0', 'epi_convert', 'epi_lds_roundtrip'), out))

    def lts_stamps(self):
        out = np.zeros(8)
        self._chk(self.lib.nbls_debug_lts_stamps(self._h, _dptr(out)))
        return dict(zip(('setup_medians', 'elemental_starts', 'csteps', 'peel', 'refine', 'finish', 'nfin', 'total'), out))

    def lts_coop_breakdown(self):
        out = np.zeros(8)
        self._chk(self.lib.nbls_debug_lts_coop_breakdown(self._h, _dptr(out)))
        return dict(zip(('groups', 'merge', 'compact', 'live_entries_total', 'w0_passes', 'w0_select', 'w0_sums', 'w0_groups'), out))

    def screen_stats(self):
        out = (C.c_int64 * 8)()
        self._chk(self.lib.nbls_debug_screen_stats(self._h, out))
        return dict(pairs=out[0], overflow=out[1], candidates=out[2], max_candidates=out[3], lag_runs=out[4],
                    lags_in_runs_of_2_or_more=out[5])

    def probe_mfma_i8(self, a, b):
        """a, b: (64, 16) int8 per-lane fragments -> (64, 4) int32 accumulators."""
        a = np.ascontiguousarray(a, dtype=np.int8).view(np.int32).reshape(64, 4)
        b = np.ascontiguousarray(b, dtype=np.int8).view(np.int32).reshape(64, 4)
        out = np.empty((64, 4), dtype=np.int32)
        self._chk(self.lib.nbls_probe_mfma_i8(self._h, _iptr(a), _iptr(b), _iptr(out)))
        return out

    def probe_mfma_f64(self, a, b):
        a = _f64(a); b = _f64(b)
        out = np.empty(256)
        self._chk(self.lib.nbls_probe_mfma_f64(self._h, _dptr(a), _dptr(b), _dptr(out)))
        return out.reshape(64, 4)
