"""Host-side planning for the HIP path: filter design, taper ramps, window plan, co-array,
and the FAST-LTS constants / start list.  Everything here is small NumPy/SciPy work done once
per call; the per-sample and per-window arithmetic runs in ``csrc/*.hip``.

Reference sources restated (file:line relative to the reference checkout; [R] = recalled
public algorithm of a dependency whose source is not in the checkout):

* filter design ............ helpers.py:126-130; obspy ``bandpass`` [R]
* taper .................... helpers.py:139; obspy ``Trace.taper`` [R]
* windows .................. lts_array ``DataBin`` [R]
* co-array, h, starts ...... lts_array ``LsBeam`` / ``LTSEstimator`` [R]
* scale factors ............ R ``robustbase::ltsReg`` (``LTScnp2``, ``LTScnp2.rew``) [R]
"""
import math

import functools

import numpy as np
from scipy import signal
from scipy.stats import norm

# ---- FAST-LTS constants [R]; one place to correct them against the real lts_array ----
LTS_N_SAMPLES = 500
LTS_CSTEPS = 4
LTS_CSTEPS2 = 100
LTS_CANDIDATES = 10
LTS_DIM = 2
LTS_QUANTILE = float(norm.ppf(0.9875))
LTS_ZERO_SCALE = 1e-7
MAD_CONST = 1.4826
TAPER_FRACTION = 0.01          # helpers.py:139


def design_bandpass(filter_type, fmin, fmax, order, ripple, fs):
    """-> (sos applied to the data, zero_phase flag, sos returned to the caller).  The design itself is
    cached (a band set is usually reused for many traces); the Nyquist warning is raised on every call."""
    sos_apply, zero_phase, sos_ret, nyquist = _design_cached(filter_type, float(fmin), float(fmax), int(order),
                                                             float(ripple), float(fs))
    if nyquist:
        import warnings
        warnings.warn('Selected high corner frequency (%s) of bandpass is at or above Nyquist (%s). '
                      'Applying a high-pass instead.' % (fmax, 0.5 * fs))
    return sos_apply.copy(), zero_phase, sos_ret.copy()


@functools.lru_cache(maxsize=4096)
def _design_cached(filter_type, fmin, fmax, order, ripple, fs):
    if filter_type == 'butter':
        fe = 0.5 * fs
        low, high = fmin / fe, fmax / fe
        nyquist = high - 1.0 > -1e-6
        if nyquist:
            z, p, k = signal.iirfilter(order, low, btype='highpass', ftype='butter', output='zpk')
            sos_apply = signal.zpk2sos(z, p, k)
            sos_ret = signal.iirfilter(order, [fmin, fmax], btype='band', ftype='butter', fs=fs, output='sos')
            return sos_apply, True, sos_ret, nyquist
        if low > 1:
            raise ValueError('Selected low corner frequency is above Nyquist.')
        # obspy designs on [fmin / (fs/2), fmax / (fs/2)] (zpk -> zpk2sos), helpers.py:128 on [fmin, fmax] with
        # fs=Fs (SciPy then forms 2*f/fs): the same floating-point numbers (scaling by two is exact), so
        # the two SOS arrays are bit-identical (tests/test_host.py checks it) and ONE design serves both
        sos = signal.iirfilter(order, [fmin, fmax], btype='band', ftype='butter', fs=fs, output='sos')
        return sos, True, sos, nyquist
    if filter_type == 'cheby1':
        sos = signal.iirfilter(order, [fmin, fmax], rp=ripple, btype='band', analog=False,
                               ftype='cheby1', fs=fs, output='sos')
        return sos, False, sos, False
    raise ValueError('unknown FILTER_TYPE %r (expected "butter" or "cheby1")' % (filter_type,))


def _bandpass_sos_batch(filter_type, f_lo, f_hi, N, rp, fs):
    """SciPy's ``iirfilter(N, [lo, hi], rp, btype='band', ftype=..., fs=fs, output='sos')`` for MANY bands at
    once: the same floating-point operations (analog prototype -> ``lp2bp_zpk`` -> ``bilinear_zpk`` ->
    ``zpk2sos`` with 'nearest' pairing), element-wise over a (bands, poles) array instead of one Python call chain
    per band (0.14 ms per band in SciPy — 7 ms of host time per 48-band call).  Band-pass only (every pole complex,
    N zeros at +1 and N at -1).  -> (B, N, 6), or None if a band does not fit the assumptions (the caller then uses
    SciPy).  tests/test_host.py compares the result bit for bit with SciPy over orders 1..8, both filter types."""
    f_lo = np.asarray(f_lo, dtype=np.float64)
    f_hi = np.asarray(f_hi, dtype=np.float64)
    B = len(f_lo)
    Wn = np.stack((f_lo, f_hi), axis=1) / (fs / 2)
    if B == 0 or np.any(Wn <= 0) or np.any(Wn >= 1) or np.any(Wn[:, 0] >= Wn[:, 1]):
        return None
    m = np.arange(-N + 1, N, 2)
    if filter_type == 'butter':
        p0 = -np.exp(1j * np.pi * m / (2 * N))
        k0 = 1
    else:
        eps = np.sqrt(10 ** (0.1 * rp) - 1.0)
        mu = 1.0 / N * np.arcsinh(1 / eps)
        theta = np.pi * m / (2 * N)
        p0 = -np.sinh(mu + 1j * theta)
        k0 = np.prod(-p0, axis=0).real
        if N % 2 == 0:
            k0 = k0 / np.sqrt(1 + eps * eps)
    fs_ = 2.0
    warped = 2 * fs_ * np.tan(np.pi * Wn / fs_)
    bw = warped[:, 1] - warped[:, 0]
    wo = np.sqrt(warped[:, 0] * warped[:, 1])
    wo2 = np.array([float(x) ** 2 for x in wo])              # Python float powers, as in lp2bp_zpk
    kbp = np.array([k0 * float(x) ** N for x in bw])
    p_lp = (p0[None, :] * bw[:, None] / 2).astype(complex)
    root = np.sqrt(p_lp ** 2 - wo2[:, None])
    p_bp = np.concatenate((p_lp + root, p_lp - root), axis=1)          # (B, 2N)
    fs2 = 2.0 * fs_
    p_z = (fs2 + p_bp) / (fs2 - p_bp)
    # gain: k * real(prod(fs2 - z) / prod(fs2 - p)); the N analog zeros sit at the origin
    num = np.prod(fs2 - np.zeros(N, dtype=complex))
    den = np.prod(fs2 - p_bp, axis=1)                                   # left-to-right along the row, as on a 1-D array
    k_z = kbp * np.real(num / den)
    # zpk2sos, 'nearest': one member of every conjugate pair, in _cplxreal's order (by real part, then |imag|)
    tol = 100 * np.finfo(float).eps
    idx = np.lexsort((np.abs(p_z.imag), p_z.real), axis=1)
    pz = np.take_along_axis(p_z, idx, axis=1)
    if np.any(np.abs(pz.imag) <= tol * np.abs(pz)):
        return None                                                     # a real pole: not the band-pass structure assumed here
    up = pz.imag > 0
    if not np.all(up.sum(axis=1) == N):
        return None
    zp = pz[up].reshape(B, N)
    zn = pz[~up].reshape(B, N)
    if N > 1 and np.any(np.diff(zp.real, axis=1) <= tol * np.abs(zp[:, :-1])):
        return None                                                     # equal real parts: _cplxreal re-sorts such runs, leave it to SciPy
    pc = (zp + zn.conj()) / 2
    # sections are filled from the last one: the pole closest to the unit circle first (argmin with removal = a
    # stable ascending sort), each with its two nearest remaining zeros (all real: N at -1, N at +1)
    order = np.argsort(np.abs(1 - np.abs(pc)), axis=1, kind='stable')
    n_minus = np.full(B, N)
    n_plus = np.full(B, N)
    sos = np.zeros((B, N, 6))
    sos[:, :, 0] = 1.0
    sos[:, :, 3] = 1.0
    rows = np.arange(B)
    for kk in range(N):
        si = N - 1 - kk
        p1 = pc[rows, order[:, kk]]
        d_minus = np.abs(-1.0 - p1)
        d_plus = np.abs(1.0 - p1)
        if np.any(d_minus == d_plus):
            return None                                                 # a tie: argsort's pick is arbitrary, leave it to SciPy
        zz = []
        for _ in range(2):
            take_minus = (n_minus > 0) & ((d_minus < d_plus) | (n_plus == 0))
            zz.append(np.where(take_minus, -1.0, 1.0))
            n_minus = n_minus - take_minus
            n_plus = n_plus - (~take_minus)
        z1, z2 = zz
        pr, pi_ = p1.real, p1.imag
        sos[:, si, 1] = (-z2) + (-z1)                                   # np.poly([z1, z2]) by convolution
        sos[:, si, 2] = (-z1) * (-z2)
        sos[:, si, 4] = (-pr) + (-pr)                                   # np.poly([p1, conj(p1)]).real
        sos[:, si, 5] = pr * pr + pi_ * pi_
    sos[:, 0, :3] *= k_z[:, None]
    return sos


def design_bandpass_many(filter_type, edges, order, ripple, fs):
    """``design_bandpass`` for a list of (fmin, fmax): cached designs are reused, the others are designed in one
    vectorised pass (``_bandpass_sos_batch``) and entered into the same cache.  -> list of (applied sos, zero_phase,
    returned sos), exactly what ``design_bandpass`` returns band by band."""
    out = [None] * len(edges)
    todo = []
    for n, (fmin, fmax) in enumerate(edges):
        key = (filter_type, float(fmin), float(fmax), int(order), float(ripple), float(fs))
        hit = _batch_cache.get(key)
        if hit is not None:
            out[n] = (hit.copy(), filter_type == 'butter', hit.copy())
        else:
            todo.append((n, key))
    if todo and filter_type in ('butter', 'cheby1'):
        fe = 0.5 * fs
        ok = [t for t in todo if (filter_type != 'butter' or not (t[1][2] / fe - 1.0 > -1e-6))]
        batch = _bandpass_sos_batch(filter_type, [t[1][1] for t in ok], [t[1][2] for t in ok], int(order), float(ripple), float(fs)) if ok else None
        if batch is not None:
            for (n, key), sos in zip(ok, batch):
                if len(_batch_cache) > 8192:
                    _batch_cache.clear()
                _batch_cache[key] = sos
                out[n] = (sos.copy(), filter_type == 'butter', sos.copy())
    for n, (fmin, fmax) in enumerate(edges):
        if out[n] is None:              # Nyquist high-pass case, unusual pole structure, unknown type (raises): band by band
            out[n] = design_bandpass(filter_type, fmin, fmax, order, ripple, fs)
    return out


_batch_cache = {}


def design_cache_clear():
    """Forget cached filter designs (bench.py clears them before every timed call)."""
    _design_cached.cache_clear()
    _batch_cache.clear()


def pad_sections(sos_list):
    """Stack per-band SOS arrays to (B, Smax, 6), padding with identity sections."""
    smax = max(s.shape[0] for s in sos_list)
    out = np.zeros((len(sos_list), smax, 6))
    out[:, :, 0] = 1.0
    out[:, :, 3] = 1.0
    for b, s in enumerate(sos_list):
        out[b, :s.shape[0], :] = s
    return out


@functools.lru_cache(maxsize=8)
def _taper_ramps_cached(npts, max_percentage):
    wlen = min(int(max_percentage * npts), int(npts / 2))
    if wlen == 0:
        return np.zeros(0), np.zeros(0)
    sides = signal.windows.hann(2 * wlen if 2 * wlen == npts else 2 * wlen + 1)
    tl, tr = np.ascontiguousarray(sides[:wlen]), np.ascontiguousarray(sides[len(sides) - wlen:])
    tl.flags.writeable = False
    tr.flags.writeable = False
    return tl, tr


def taper_ramps(npts, max_percentage=TAPER_FRACTION):
    """Left and right ramps of obspy's Hann taper (ones in between are implicit).  A pure function of the trace
    length: kept for the last few lengths (read-only arrays)."""
    return _taper_ramps_cached(int(npts), float(max_percentage))


def window_plan(npts, fs, window_length, window_overlap):
    """(W, inc, number of windows): W = int(winlen*fs), inc = int(round((1-ov)*W)),
    windows start at arange(0, npts - W, inc)."""
    W = int(window_length * fs)
    inc = int(np.round((1 - window_overlap) * W))
    if W < 2 or inc < 1:
        raise ValueError('window length / overlap give an empty window or zero hop')
    nwin = max(0, -(-(int(npts) - W) // inc))          # = len(np.arange(0, npts - W, inc)), without building it
    return W, inc, nwin


def pair_table(nchans):
    return np.array([(i, j) for i in range(nchans - 1) for j in range(i + 1, nchans)], dtype=np.int32)


@functools.lru_cache(maxsize=8)
def _co_array_cached(shape, raw):
    out = _co_array(np.frombuffer(raw, dtype=np.float64).reshape(shape))
    for a in out:
        a.flags.writeable = False
    return out


def co_array(rij):
    """rij (2, N) km -> xij (P, 2) = r_i - r_j, pair table (P, 2), pinv(xij) (2, P).  A pure function of the
    coordinates: the last few geometries are kept (read-only arrays; ``functools.lru_cache``: safe under the host threads
    that drive several handles — the dict with manual eviction it replaces was not)."""
    rij = np.ascontiguousarray(rij, dtype=np.float64)
    return _co_array_cached(rij.shape, rij.tobytes())


def _co_array(rij):
    idx = pair_table(rij.shape[1])
    xij = np.ascontiguousarray((rij[:, idx[:, 0]] - rij[:, idx[:, 1]]).T)
    if np.linalg.matrix_rank(xij) < LTS_DIM:
        raise RuntimeError('Co-array is ill posed for the least squares problem. Check array coordinates.')
    return xij, idx, np.ascontiguousarray(np.linalg.pinv(xij))


def lts_h(P, alpha, p=LTS_DIM):
    n2 = (P + p + 1) // 2
    return int(math.floor(2 * n2 - P + 2 * (P - n2) * alpha))


@functools.lru_cache(maxsize=64)
def _uniran_cached(P, n_samples, p):
    seed = 0
    out = np.empty((n_samples, p), dtype=np.int64)
    for s in range(n_samples):
        chosen = []
        for _ in range(p):
            while True:
                seed = (seed * 5761 + 999) % 65536
                num = int(math.floor(seed / 65536.0 * P))
                if num not in chosen:
                    break
            chosen.append(num)
        out[s] = chosen
    out.flags.writeable = False
    return out


def uniran_subsets(P, n_samples=LTS_N_SAMPLES, p=LTS_DIM):
    """robustbase LCG (seed*5761+999 mod 65536), seed 0, carried across subsets.  A pure function of
    the pair count (the sequence is generated once per P and kept)."""
    return _uniran_cached(int(P), int(n_samples), int(p)).copy()


@functools.lru_cache(maxsize=64)
def _all_pairs(P):
    out = np.array([(i, j) for i in range(P - 1) for j in range(i + 1, P)], dtype=np.int64)
    out.flags.writeable = False
    return out


def lts_starts(xs):
    """Elemental starts on the standardised co-array: every 2-subset if there are at most
    LTS_N_SAMPLES of them, else LTS_N_SAMPLES LCG-random ones; rank-deficient subsets are
    extended until they have rank 2.  (S, 4) int32, -1 padded."""
    P = xs.shape[0]
    if P * (P - 1) // 2 <= LTS_N_SAMPLES:
        subs = _all_pairs(int(P))
    else:
        subs = uniran_subsets(P)
    out = -np.ones((len(subs), 4), dtype=np.int32)
    out[:, :2] = subs
    scale = np.max(np.abs(xs)) ** 2
    # Gram determinant of every 2-subset at once; only the (rare) rank-deficient ones take the repair loop
    a, b = xs[subs[:, 0]], xs[subs[:, 1]]
    g00 = a[:, 0] * a[:, 0] + b[:, 0] * b[:, 0]
    g01 = a[:, 0] * a[:, 1] + b[:, 0] * b[:, 1]
    g11 = a[:, 1] * a[:, 1] + b[:, 1] * b[:, 1]
    suspect = np.nonzero(~(g00 * g11 - g01 * g01 > 1e-9 * scale * scale))[0]      # generous margin over the 1e-12 test below
    for s in suspect.tolist():
        sub = list(subs[s])
        nxt = 0
        while True:
            g = xs[sub].T @ xs[sub]
            det = g[0, 0] * g[1, 1] - g[0, 1] * g[1, 0]
            if det > 1e-12 * scale * scale or len(sub) >= 4:
                break
            while nxt in sub:
                nxt += 1
            if nxt >= P:
                break
            sub.append(nxt)
        out[s, :len(sub)] = sub
    return out


def _consfactor(m, n):
    if m >= n or m <= 0:
        return 1.0
    q = norm.ppf((m + n) / (2.0 * n))
    return 1.0 / math.sqrt(1.0 - (2.0 * n) / (m / q) * norm.pdf(q))


def _consfactor_table(n):
    """_consfactor(m, n) for m = 1..n-1 in one vectorised pass (same IEEE operations per element)."""
    m = np.arange(1, n, dtype=np.float64)
    q = norm.ppf((m + n) / (2.0 * n))
    return 1.0 / np.sqrt(1.0 - (2.0 * n) / (m / q) * norm.pdf(q))


def _cnp2(p, n, alpha, c500, c875):
    c500 = np.asarray(c500, dtype=float)
    c875 = np.asarray(c875, dtype=float)
    y500 = np.log(-c500[0] / p ** c500[1])
    y875 = np.log(-c875[0] / p ** c875[1])
    k500 = np.linalg.solve(np.column_stack((np.ones(2), -np.log(c500[2] * p ** 2))), y500)
    k875 = np.linalg.solve(np.column_stack((np.ones(2), -np.log(c875[2] * p ** 2))), y875)
    fp500 = 1 - math.exp(k500[0]) / n ** k500[1]
    fp875 = 1 - math.exp(k875[0]) / n ** k875[1]
    if alpha <= 0.875:
        fp = fp500 + (fp875 - fp500) / 0.375 * (alpha - 0.5)
    else:
        fp = fp875 + (1 - fp875) / 0.125 * (alpha - 0.875)
    return 1.0 / fp


# robustbase LTScnp2 / LTScnp2.rew coefficient tables, no-intercept, p >= 2 [R]
_RAW_500 = [[-0.487338281979106, -0.340762058011], [0.405511279418594, 0.37972360544988], [3, 5]]
_RAW_875 = [[-0.251778730491252, -0.146660023184295], [0.883966931611758, 0.86292940340761], [3, 5]]
_REW_500 = [[-0.417574780492848, -0.175753709374146], [1.83958876341367, 1.8313809497999], [3, 5]]
_REW_875 = [[-0.267522855927958, -0.161200683014406], [1.17559984533974, 1.21675019853961], [3, 5]]


@functools.lru_cache(maxsize=64)
def _lts_factors(P, alpha):
    """Consistency and small-sample factors of the raw and the reweighted scale: pure functions of the pair count and
    alpha (like the LCG subsets above), computed once per (P, alpha) and kept."""
    h = lts_h(P, alpha)
    raw = _consfactor(h, P) * _cnp2(LTS_DIM, P, alpha, _RAW_500, _RAW_875)
    rew = np.ones(P + 1)
    cor = _cnp2(LTS_DIM, P, alpha, _REW_500, _REW_875)
    rew[1:P] = _consfactor_table(P) * cor
    rew.flags.writeable = False
    return raw, rew


def lts_plan(xij, alpha):
    """Everything the LTS kernel needs besides the lags (see ``nbls_lts_params``)."""
    if not (0.5 <= alpha < 1.0):
        raise ValueError('LTS needs 0.5 <= ALPHA < 1.0')
    P = xij.shape[0]
    h = lts_h(P, alpha)
    xij_mad = MAD_CONST * np.median(np.abs(xij), axis=0)
    if not np.all(xij_mad > 0):
        raise RuntimeError('Co-array MAD is zero along an axis; cannot standardise for LTS.')
    starts = lts_starts(xij / xij_mad)
    raw, rew = _lts_factors(int(P), float(alpha))
    rew = rew.copy()
    return dict(alpha=alpha, h=h, starts=starts, csteps=LTS_CSTEPS, csteps2=LTS_CSTEPS2,
                ncand=LTS_CANDIDATES, xij_mad=xij_mad, raw_factor=raw, rew_table=rew,
                quantile=LTS_QUANTILE, zero_scale=LTS_ZERO_SCALE)


def sosfreqz_bands(sos_list, worN, fs):
    """``scipy.signal.sosfreqz(sos, worN, fs=fs)`` for every band of a call (narrow_band_least_squares.py:78-80) ->
    (w (F,), h rows (B, F) complex).  The same NumPy operations in the same order on arrays of the same shape as
    SciPy's ``freqz_sos`` -> ``freqz`` -> ``numpy.polynomial.polynomial.polyval`` chain (``zm1 = exp(-1j w)``; Horner in
    ``zm1`` per section; ``h *= num / den``), so every value has SciPy's bits — but ``zm1`` is computed once per call
    instead of once per section and band, and the per-call argument checks are gone (48 bands: 2.7 -> 1.2 ms).
    Falls back to SciPy for anything but an array of frequencies and (S, 6) float sections."""
    if worN is None or np.ndim(worN) == 0:          # a point count (or SciPy's default): its own branches
        return None
    w_in = np.atleast_1d(worN)
    if w_in.ndim != 1 or w_in.dtype.kind not in 'fiu' or not (np.isscalar(fs) and fs > 0):
        return None
    w = 2 * np.pi * w_in / fs
    zm1 = np.exp(-1j * w)
    x0 = zm1 * 0
    rows = np.empty((len(sos_list), len(w)), dtype=complex)
    for n, sos in enumerate(sos_list):
        sos = np.asarray(sos)
        if sos.ndim != 2 or sos.shape[1] != 6 or sos.shape[0] == 0 or sos.dtype != np.float64:
            return None
        h = 1.
        for row in sos:
            num = row[2] + x0
            num = row[1] + num * zm1
            num = row[0] + num * zm1
            den = row[5] + x0
            den = row[4] + den * zm1
            den = row[3] + den * zm1
            h *= num / den
        rows[n] = h
    return w * (fs / (2 * np.pi)), rows


def uncertainty_frame(xij):
    """(lambda_0, lambda_1, R00, R01, R10, R11): eigenvalues (ascending, ``numpy.linalg.eigh``) of X^T X of the co-array
    and the rotation into its eigen-frame as lts_array builds it [R: source absent] — ``ang = arccos(evecs[0, 0])``,
    ``R = [[cos, sin], [-sin, cos]]`` — for the confidence intervals the GPU computes behind the solve
    (``nbls_set_uncertainty``; 7th / 8th returns of ``ltsva``, narrow_band_least_squares.py:91)."""
    xij = np.asarray(xij, dtype=np.float64)
    evals, evecs = np.linalg.eigh(xij.T @ xij)
    ang = np.arccos(np.clip(evecs[0, 0], -1.0, 1.0))
    return np.array([evals[0], evals[1], np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)])
