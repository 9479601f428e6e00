"""CPU oracle for the narrow-band least-squares / LTS hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The shipped package (``narrow_band_least_squares_amd``) never imports
anything under ``oracle/`` and fails loudly when its HIP library is missing.

It is a NumPy/SciPy restatement of the reference's algorithm for the path

    filter_data -> ltsva (window, pairwise xcorr, lag pick, MdCCM, OLS | FAST-LTS)
                -> narrow_band_least_squares packing

Reference citations (``file:line`` relative to /root/reference):

* band loop / packing .......... narrow_band_least_squares.py:41-127, 134-218, 259-323
* filter + taper ............... helpers.py:108-141
* geometry ..................... helpers.py:239-284
* ``ltsva`` .................... call sites narrow_band_least_squares.py:91,183;
                                 example.py:109.  THE SOURCE OF ``lts_array`` IS NOT IN
                                 THE CONTAINER (empty git submodule, .gitmodules:1-3, no
                                 pinned SHA).  Everything inside ``ltsva`` below restates
                                 the published algorithm of uafgeotools/lts_array
                                 (Bishop et al. 2020, GJI) and R ``robustbase::ltsReg``
                                 (Rousseeuw & Van Driessen 2006) as summarised in
                                 SURVEY.md §3.3/§8a.
* obspy ``bandpass(zerophase=True)`` / ``Trace.taper`` — obspy is absent too; restated
  from the published recipe (SURVEY.md §8 row a3').

PARITY STATUS: the sub-steps that delegate to NumPy/SciPy (iirfilter, sosfilt,
zpk2sos, np.correlate, argmax, nanmedian, lstsq) are pinned against those libraries
and the packing is pinned against the reference's own outer loop (tests/golden).
Everything that lives only inside lts_array/obspy (FAST-LTS constants, reweighting
factors, sigma_tau under LTS, stdict key format, Vincenty) is **parity unpinned**
against the real third-party code: the reference ships no fixture for it.

Arithmetic-order contract with the HIP kernels (so that discrete LTS decisions
agree bit for bit): every 2x2 least-squares fit is solved from normal equations
accumulated sequentially over the pair index k = 0..P-1 in plain IEEE double
(no FMA contraction), by Cramer's rule; the h-subset is chosen by stable rank of
|r_k| (ties -> lower k first); the objective is the sum of r_k^2 over the subset in
ascending k.  ``fast_lts_literal`` keeps the lstsq/argsort form of the published
algorithm and is used by the tests to show both forms agree.
"""
import math

import numpy as np
from scipy import signal
from scipy.stats import norm

# --------------------------------------------------------------------------
# LTS constants [R: lts_array LTSEstimator.__init__ / robustbase ltsReg]
# centralised so that a maintainer with the real lts_array can correct them.
# --------------------------------------------------------------------------
LTS_N_SAMPLES = 500        # number of random starts
LTS_CSTEPS = 4             # C-steps per start
LTS_CSTEPS2 = 100          # max C-steps in the final refinement
LTS_CANDIDATES = 10        # best candidates kept for refinement
LTS_DIM = 2                # unknowns (2-D slowness)
LTS_QUANTILE = float(norm.ppf(0.9875))   # robustbase "quantiel"
LTS_ZERO_SCALE = 1e-7      # robustbase: |s0| < 1e-7 -> exact-fit branch
MAD_CONST = 1.4826


# --------------------------------------------------------------------------
# tiny duck-typed stream (what the reference touches: SURVEY.md §8b)
# --------------------------------------------------------------------------
class OStats:
    def __init__(self, sampling_rate, npts, starttime=0.0, latitude=None, longitude=None):
        self.sampling_rate = float(sampling_rate)
        self.npts = int(npts)
        self.starttime = starttime
        self.latitude = latitude
        self.longitude = longitude


class OTrace:
    def __init__(self, data, stats):
        self.data = np.asarray(data)
        self.stats = stats


class OStream(list):
    def copy(self):
        return OStream(OTrace(tr.data.copy(), OStats(tr.stats.sampling_rate, tr.stats.npts,
                                                     tr.stats.starttime, tr.stats.latitude,
                                                     tr.stats.longitude)) for tr in self)


def make_stream(data, fs, starttime=0.0, lat=None, lon=None):
    """data: (N, npts) array -> OStream."""
    data = np.asarray(data, dtype=np.float64)
    st = OStream()
    for i in range(data.shape[0]):
        st.append(OTrace(data[i].copy(), OStats(fs, data.shape[1], starttime,
                                                None if lat is None else lat[i],
                                                None if lon is None else lon[i])))
    return st


def start_datenum(starttime):
    """matplotlib date number (days since 1970-01-01, mpl >= 3.3 epoch) of a start time."""
    if hasattr(starttime, 'matplotlib_date'):
        return float(starttime.matplotlib_date)
    if hasattr(starttime, 'timestamp') and not isinstance(starttime, (int, float)):
        ts = starttime.timestamp
        ts = ts() if callable(ts) else ts
        return float(ts) / 86400.0
    if isinstance(starttime, np.datetime64):
        return float((starttime - np.datetime64('1970-01-01T00:00:00')) / np.timedelta64(1, 'us')) / 86400e6
    return float(starttime)


def times_matplotlib(tr):
    """obspy Trace.times('matplotlib') [R]: start date number + n/fs seconds in days."""
    n = len(tr.data)
    return start_datenum(tr.stats.starttime) + (np.arange(n) / tr.stats.sampling_rate) / 86400.0


# --------------------------------------------------------------------------
# helpers.py restatements
# --------------------------------------------------------------------------
def make_float(x):
    """helpers.py:145-158."""
    return np.array([float(x[jj]) for jj in range(len(x))])


def design_bandpass(filter_type, fmin, fmax, order, ripple, fs):
    """SOS actually applied to the data + zero-phase flag + SOS handed back to the caller.

    helpers.py:126-130.  'butter' applies obspy's bandpass [R: obspy/signal/filter.py]:
    zpk design on frequencies normalised by Nyquist, zpk2sos, forward-backward; the SOS
    returned to the caller is the separate ``iirfilter(..., fs=Fs, output='sos')`` of
    helpers.py:128.  'cheby1' applies the returned SOS causally (helpers.py:130-137).
    """
    if filter_type == 'butter':
        fe = 0.5 * fs
        low = fmin / fe
        high = fmax / fe
        if high - 1.0 > -1e-6:
            # obspy: "Selected high corner frequency is above Nyquist. Applying a high-pass instead."
            z, p, k = signal.iirfilter(order, low, btype='highpass', ftype='butter', output='zpk')
        else:
            if low > 1:
                raise ValueError('Selected low corner frequency is above Nyquist.')
            z, p, k = signal.iirfilter(order, [low, high], btype='band', ftype='butter', output='zpk')
        sos_apply = signal.zpk2sos(z, p, k)
        sos_ret = signal.iirfilter(order, [fmin, fmax], btype='band', ftype='butter', fs=fs, output='sos')
        return sos_apply, True, sos_ret
    elif filter_type == 'cheby1':
        sos = signal.iirfilter(order, [fmin, fmax], rp=ripple, btype='band', analog=False,
                               ftype='cheby1', fs=fs, output='sos')
        return sos, False, sos
    raise ValueError('unknown FILTER_TYPE %r' % (filter_type,))


def taper_window(npts, max_percentage=0.01):
    """obspy Trace.taper(max_percentage, type='hann', side='both') [R: obspy/core/trace.py]."""
    wlen = min(int(max_percentage * npts), int(npts / 2))
    if 2 * wlen == npts:
        sides = signal.windows.hann(2 * wlen)
    else:
        sides = signal.windows.hann(2 * wlen + 1)
    return np.hstack((sides[:wlen], np.ones(npts - 2 * wlen), sides[len(sides) - wlen:]))


def filter_data(st, filter_type, fmin, fmax, order, ripple):
    """helpers.py:108-141 -> (stf, Fs, sos)."""
    stf = st.copy()
    fs = stf[0].stats.sampling_rate
    sos_apply, zero_phase, sos_ret = design_bandpass(filter_type, fmin, fmax, order, ripple, fs)
    for ii in range(len(st)):
        x = np.asarray(stf[ii].data, dtype=np.float64)
        if zero_phase:
            first = signal.sosfilt(sos_apply, x)
            y = signal.sosfilt(sos_apply, first[::-1])[::-1]
        else:
            y = signal.sosfilt(sos_apply, x)
        stf[ii].data = y
    for tr in stf:
        tr.data = tr.data * taper_window(len(tr.data), 0.01)
    return stf, fs, sos_ret


def vincenty_inverse(lat1, lon1, lat2, lon2):
    """Vincenty (1975) inverse on WGS84 -> (distance m, azimuth 1->2 deg, azimuth 2->1 deg).

    Stands in for obspy.geodetics.base.calc_vincenty_inverse (helpers.py:4,271); obspy is
    absent, so this follows the published formulae.
    """
    a = 6378137.0
    f = 1.0 / 298.257223563
    b = (1.0 - f) * a
    if lat1 == lat2 and lon1 == lon2:
        return 0.0, 0.0, 0.0
    p1, p2 = math.radians(lat1), math.radians(lat2)
    L = math.radians(lon2 - lon1)
    # keep L in (-pi, pi]
    L = (L + math.pi) % (2 * math.pi) - math.pi
    U1 = math.atan((1 - f) * math.tan(p1))
    U2 = math.atan((1 - f) * math.tan(p2))
    sU1, cU1, sU2, cU2 = math.sin(U1), math.cos(U1), math.sin(U2), math.cos(U2)
    lam = L
    for _ in range(200):
        sl, cl = math.sin(lam), math.cos(lam)
        sin_sig = math.sqrt((cU2 * sl) ** 2 + (cU1 * sU2 - sU1 * cU2 * cl) ** 2)
        if sin_sig == 0.0:
            return 0.0, 0.0, 0.0
        cos_sig = sU1 * sU2 + cU1 * cU2 * cl
        sig = math.atan2(sin_sig, cos_sig)
        sin_al = cU1 * cU2 * sl / sin_sig
        cos2_al = 1.0 - sin_al * sin_al
        cos2sm = cos_sig - 2.0 * sU1 * sU2 / cos2_al if cos2_al != 0.0 else 0.0
        C = f / 16.0 * cos2_al * (4.0 + f * (4.0 - 3.0 * cos2_al))
        lam_new = L + (1.0 - C) * f * sin_al * (
            sig + C * sin_sig * (cos2sm + C * cos_sig * (-1.0 + 2.0 * cos2sm * cos2sm)))
        done = abs(lam_new - lam) < 1e-12
        lam = lam_new
        if done:
            break
    sl, cl = math.sin(lam), math.cos(lam)
    u2 = cos2_al * (a * a - b * b) / (b * b)
    A = 1.0 + u2 / 16384.0 * (4096.0 + u2 * (-768.0 + u2 * (320.0 - 175.0 * u2)))
    B = u2 / 1024.0 * (256.0 + u2 * (-128.0 + u2 * (74.0 - 47.0 * u2)))
    dsig = B * sin_sig * (cos2sm + B / 4.0 * (
        cos_sig * (-1.0 + 2.0 * cos2sm ** 2)
        - B / 6.0 * cos2sm * (-3.0 + 4.0 * sin_sig ** 2) * (-3.0 + 4.0 * cos2sm ** 2)))
    dist = b * A * (sig - dsig)
    az12 = math.degrees(math.atan2(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl)) % 360.0
    az21 = (math.degrees(math.atan2(cU1 * sl, -sU1 * cU2 + cU1 * sU2 * cl)) + 180.0) % 360.0
    return dist, az12, az21


def get_rij(latlist, lonlist, nchans):
    """helpers.py:239-284."""
    if (len(latlist) != nchans) or (len(lonlist) != nchans):
        raise ValueError('Mismatch between the number of stream channels and the latitude or longitude list length.')
    xnew = np.zeros((nchans,))
    ynew = np.zeros((nchans,))
    for jj in range(1, nchans):
        delta, az, _ = vincenty_inverse(latlist[0], lonlist[0], latlist[jj], lonlist[jj])
        az = (450 - az) % 360
        xnew[jj] = delta / 1000 * np.cos(az * np.pi / 180)
        ynew[jj] = delta / 1000 * np.sin(az * np.pi / 180)
    xnew -= np.mean(xnew)
    ynew -= np.mean(ynew)
    return np.array([xnew.tolist(), ynew.tolist()])


# --------------------------------------------------------------------------
# lts_array restatement [R]
# --------------------------------------------------------------------------
def window_plan(npts, fs, window_length, window_overlap):
    """DataBin [R]: W, inc, window start indices."""
    W = int(window_length * fs)
    inc = int(np.round((1 - window_overlap) * W))
    intervals = np.arange(0, npts - W, inc, dtype='int')
    return W, inc, intervals


def pair_table(nchans):
    """idx_pair = [(i, j) for i < j] [R]."""
    return [(i, j) for i in range(nchans - 1) for j in range(i + 1, nchans)]


def co_array(rij):
    """xij[k] = rij[:, i] - rij[:, j] [R]."""
    idx = pair_table(rij.shape[1])
    return np.array([rij[:, i] - rij[:, j] for (i, j) in idx]), idx


def correlate_windows(data, W, intervals, idx_pair, fs):
    """LsBeam.correlate [R].  data: (npts, N).  -> tau (P, nits), mdccm (nits), cmax (P, nits)."""
    nits = len(intervals)
    P = len(idx_pair)
    tau = np.empty((P, nits))
    mdccm = np.full(nits, np.nan)
    cmax_all = np.empty((P, nits))
    with np.errstate(invalid='ignore', divide='ignore'):
        for jj in range(nits):
            t0 = intervals[jj]
            tf = t0 + W
            cij = np.empty((2 * W - 1, P))
            for k, (i, j) in enumerate(idx_pair):
                a = data[t0:tf, i]
                b = data[t0:tf, j]
                cij[:, k] = np.correlate(a, b, mode='full') / np.sqrt(np.sum(a * a) * np.sum(b * b))
            cmax = cij.max(axis=0)
            cmax_all[:, jj] = cmax
            if np.all(np.isnan(cmax)):
                mdccm[jj] = np.nan
            else:
                mdccm[jj] = np.nanmedian(cmax)
            delay = np.argmax(cij, axis=0) + 1
            tau[:, jj] = (W - delay) / fs
    return tau, mdccm, cmax_all


def vel_baz(z):
    """vel = 1/||z||, baz = (atan2(z0, z1) deg - 360) % 360 [R]."""
    with np.errstate(divide='ignore', invalid='ignore'):
        vel = 1.0 / np.sqrt(z[0] * z[0] + z[1] * z[1])
        baz = (np.arctan2(z[0], z[1]) * 180.0 / np.pi - 360.0) % 360.0
    return vel, baz


def ols_solve(xij, tau):
    """OLSEstimator.solve [R].  z = pinv(X) tau; sigma_tau = sqrt(tau.r/(P-2))."""
    P = xij.shape[0]
    xpinv = np.linalg.pinv(xij)                    # (2, P), pre-computed once
    nits = tau.shape[1]
    z = np.zeros((2, nits))
    for k in range(P):                              # sequential k accumulation (order contract)
        z[0] += xpinv[0, k] * tau[k]
        z[1] += xpinv[1, k] * tau[k]
    acc = np.zeros(nits)
    for k in range(P):
        r = tau[k] - (xij[k, 0] * z[0] + xij[k, 1] * z[1])
        acc += tau[k] * r
    with np.errstate(invalid='ignore'):
        sigma_tau = np.sqrt(acc / (P - LTS_DIM))
    vel, baz = vel_baz(z)
    return z, vel, baz, sigma_tau


def lts_h(P, alpha, p=LTS_DIM):
    """robustbase h.alpha.n [R]."""
    n2 = (P + p + 1) // 2
    return int(math.floor(2 * n2 - P + 2 * (P - n2) * alpha))


def uniran_subsets(P, n_samples=LTS_N_SAMPLES, p=LTS_DIM):
    """random_set/uniran [R]: robustbase LCG, seed carried across subsets, starting at 0."""
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
    return out


def lts_starts(xij_std):
    """Start list: all C(P,2) 2-subsets if C(P,2) <= n_samples else 500 LCG subsets.

    Rank-deficient 2-subsets (collinear co-array vectors) get further points appended until the
    subset has rank 2 ("add points until full rank" [R]).  Returns int array (S, 4), -1 padded.
    """
    P = xij_std.shape[0]
    ncomb = P * (P - 1) // 2
    if ncomb <= LTS_N_SAMPLES:
        subs = [(i, j) for i in range(P - 1) for j in range(i + 1, P)]
    else:
        subs = [tuple(r) for r in uniran_subsets(P)]
    out = -np.ones((len(subs), 4), dtype=np.int32)
    scale = np.max(np.abs(xij_std)) ** 2
    for s, sub in enumerate(subs):
        sub = list(sub)
        nxt = 0
        while True:
            xs = xij_std[sub]
            g = xs.T @ xs
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


def _solve_normal(sxx, sxy, syy, bx, by):
    """Cramer's rule for [[sxx,sxy],[sxy,syy]] z = [bx,by] (order contract with the HIP kernel)."""
    det = sxx * syy - sxy * sxy
    with np.errstate(divide='ignore', invalid='ignore'):
        z0 = (bx * syy - by * sxy) / det
        z1 = (by * sxx - bx * sxy) / det
    return z0, z1


def _fit_masked(X, y, mask):
    """LS fit on the pairs selected by mask.  X: (P,2); y: (..., P); mask: (..., P) bool.
    Sequential accumulation over k with masked-out terms skipped."""
    shp = mask.shape[:-1]
    sxx = np.zeros(shp); sxy = np.zeros(shp); syy = np.zeros(shp)
    bx = np.zeros(shp); by = np.zeros(shp)
    P = X.shape[0]
    for k in range(P):
        m = mask[..., k]
        yk = y[..., k]
        sxx = np.where(m, sxx + X[k, 0] * X[k, 0], sxx)
        sxy = np.where(m, sxy + X[k, 0] * X[k, 1], sxy)
        syy = np.where(m, syy + X[k, 1] * X[k, 1], syy)
        bx = np.where(m, bx + X[k, 0] * yk, bx)
        by = np.where(m, by + X[k, 1] * yk, by)
    return _solve_normal(sxx, sxy, syy, bx, by)


def _residuals(X, y, z0, z1):
    """r_k = (y_k - x_k0 z0) - x_k1 z1 (this exact association is part of the order contract)."""
    return (y - X[:, 0] * z0[..., None]) - X[:, 1] * z1[..., None]


def _select(X, y, z0, z1, h):
    """h-subset of the current fit: stable rank of |r_k| < h; objective = sum of r_k^2 over it
    in ascending k."""
    r = _residuals(X, y, z0, z1)
    order = np.argsort(np.abs(r), axis=-1, kind='stable')
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.arange(r.shape[-1]), axis=-1)
    mask = rank < h
    obj = np.zeros(r.shape[:-1])
    for k in range(r.shape[-1]):
        obj = np.where(mask[..., k], obj + r[..., k] * r[..., k], obj)
    return mask, obj


def _cstep(X, y, mask, h):
    """One C-step [R: cstep]: refit on the current h-subset, then re-select.  The objective is the
    sum of the h smallest squared residuals of the NEW fit; its subset seeds the next C-step."""
    z0, z1 = _fit_masked(X, y, mask)
    mask, obj = _select(X, y, z0, z1, h)
    return z0, z1, mask, obj


def fast_lts(tau, xij, alpha, starts=None):
    """fast_LTS [R] for all windows of one band.  tau: (P, nits) -> z (2, nits) (NaN where MAD(tau)==0).

    Vectorised over (window, start); arithmetic order as in the module docstring.
    """
    P, nits = tau.shape
    h = lts_h(P, alpha)
    xij_mad = MAD_CONST * np.median(np.abs(xij), axis=0)
    X = xij / xij_mad
    if starts is None:
        starts = lts_starts(X)
    S = starts.shape[0]
    tmad = MAD_CONST * np.median(np.abs(tau), axis=0)          # (nits,)
    live = tmad != 0
    with np.errstate(divide='ignore', invalid='ignore'):
        y = (tau / tmad).T                                      # (nits, P)
    y = np.where(live[:, None], y, 0.0)
    # initial exact fit on each start subset
    smask = np.zeros((S, P), dtype=bool)
    for s in range(S):
        smask[s, starts[s][starts[s] >= 0]] = True
    yb = np.broadcast_to(y[:, None, :], (nits, S, P))
    mb = np.broadcast_to(smask[None, :, :], (nits, S, P))
    z0, z1 = _fit_masked(X, yb, mb)
    mask, _ = _select(X, yb, z0, z1, h)
    obj = np.full((nits, S), np.inf)
    prev = np.zeros((nits, S))
    active = np.ones((nits, S), dtype=bool)
    for kk in range(LTS_CSTEPS):
        z0n, z1n, maskn, objn = _cstep(X, yb, mask, h)
        z0 = np.where(active, z0n, z0)
        z1 = np.where(active, z1n, z1)
        obj = np.where(active, objn, obj)
        mask = np.where(active[..., None], maskn, mask)
        if kk >= 1:
            active = active & ~(obj == prev)
        prev = np.where(active, obj, prev)
    # keep the LTS_CANDIDATES best distinct (obj, z) per window, then refine
    zout = np.full((2, nits), np.nan)
    for jj in range(nits):
        if not live[jj]:
            continue
        o = obj[jj]
        key = np.where(np.isnan(o), np.inf, o)
        order = np.argsort(key, kind='stable')
        cand = []
        for s in order:
            if not np.isfinite(key[s]):
                break
            dup = False
            for c in cand:
                if o[c] == o[s] and z0[jj, c] == z0[jj, s] and z1[jj, c] == z1[jj, s]:
                    dup = True
                    break
            if not dup:
                cand.append(s)
                if len(cand) == LTS_CANDIDATES:
                    break
        best = np.inf
        bz = (np.nan, np.nan)
        for c in cand:
            c0 = np.array(z0[jj, c]); c1 = np.array(z1[jj, c])
            cmask, _ = _select(X, y[jj], c0, c1, h)
            pobj = 0.0
            cobj = np.inf
            for kk in range(LTS_CSTEPS2):
                c0, c1, cmask, cobj = _cstep(X, y[jj], cmask, h)
                if kk >= 1 and cobj == pobj:
                    break
                pobj = cobj
            if cobj < best:
                best = float(cobj)
                bz = (float(c0), float(c1))
        zout[0, jj] = bz[0] * tmad[jj] / xij_mad[0]
        zout[1, jj] = bz[1] * tmad[jj] / xij_mad[1]
    return zout


def fast_lts_literal(tau_col, xij, alpha):
    """The same algorithm for ONE window in the literal lstsq/argsort form of the published code
    (slow; used by tests to cross-check ``fast_lts``).  Returns z (2,) or NaNs."""
    P = len(tau_col)
    h = lts_h(P, alpha)
    xij_mad = MAD_CONST * np.median(np.abs(xij), axis=0)
    X = xij / xij_mad
    tmad = MAD_CONST * np.median(np.abs(tau_col))
    if tmad == 0:
        return np.array([np.nan, np.nan])
    y = tau_col / tmad
    starts = lts_starts(X)
    objs, coefs = [], []

    def cstep(z):
        r = y - X @ z
        idx = np.sort(np.argsort(np.abs(r), kind='stable')[:h])
        z = np.linalg.lstsq(X[idx], y[idx], rcond=None)[0]
        r = y - X @ z
        return z, float(np.sum(np.sort(np.abs(r))[:h] ** 2))

    for s in range(starts.shape[0]):
        sub = starts[s][starts[s] >= 0]
        z = np.linalg.lstsq(X[sub], y[sub], rcond=None)[0]
        prev = 0.0
        obj = np.inf
        for kk in range(LTS_CSTEPS):
            z, obj = cstep(z)
            if kk >= 1 and obj == prev:
                break
            prev = obj
        objs.append(obj); coefs.append(z)
    order = np.argsort(np.array(objs), kind='stable')[:4 * LTS_CANDIDATES]
    best, bz = np.inf, np.array([np.nan, np.nan])
    kept = []
    for s in order:
        if any(abs(objs[s] - objs[c]) <= 1e-12 * max(1.0, objs[c]) and
               np.allclose(coefs[s], coefs[c], rtol=1e-9, atol=1e-12) for c in kept):
            continue
        kept.append(s)
        if len(kept) > LTS_CANDIDATES:
            break
        z = coefs[s]
        prev = 0.0
        obj = np.inf
        for kk in range(LTS_CSTEPS2):
            z, obj = cstep(z)
            if kk >= 1 and obj == prev:
                break
            prev = obj
        if obj < best:
            best, bz = obj, z
    return bz * tmad / xij_mad


# ---- robustbase small-sample / consistency factors [R] -----------------------------------
def raw_consfactor(h, n):
    if h >= n:
        return 1.0
    q = norm.ppf((h + n) / (2.0 * n))
    return 1.0 / math.sqrt(1.0 - (2.0 * n) / (h / q) * norm.pdf(q))


def _cnp2(p, n, alpha, c500, c875):
    """Shared body of robustbase LTScnp2 / LTScnp2.rew for p >= 2, intercept = FALSE."""
    c500 = np.asarray(c500, dtype=float)   # rows: alfaq, betaq, qwaarden ; cols: q=3, q=5
    c875 = np.asarray(c875, dtype=float)
    y500 = np.log(-c500[0] / p ** c500[1])
    y875 = np.log(-c875[0] / p ** c875[1])
    A500 = np.column_stack((np.ones(2), -np.log(c500[2] * p ** 2)))
    A875 = np.column_stack((np.ones(2), -np.log(c875[2] * p ** 2)))
    k500 = np.linalg.solve(A500, y500)
    k875 = np.linalg.solve(A875, y875)
    fp500 = 1 - math.exp(k500[0]) / n ** k500[1]
    fp875 = 1 - math.exp(k875[0]) / n ** k875[1]
    if alpha <= 0.875:
        fp = fp500 + (fp875 - fp500) / 0.375 * (alpha - 0.5)
    else:
        fp = fp875 + (1 - fp875) / 0.125 * (alpha - 0.875)
    return 1.0 / fp


def raw_corfactor(p, n, alpha):
    """robustbase LTScnp2(p, intercept=FALSE, n, alpha), p >= 2 branch [R: constants recalled]."""
    return _cnp2(p, n, alpha,
                 [[-0.487338281979106, -0.340762058011], [0.405511279418594, 0.37972360544988], [3, 5]],
                 [[-0.251778730491252, -0.146660023184295], [0.883966931611758, 0.86292940340761], [3, 5]])


def rew_corfactor(p, n, alpha):
    """robustbase LTScnp2.rew(p, intercept=FALSE, n, alpha), p >= 2 branch [R: constants recalled]."""
    return _cnp2(p, n, alpha,
                 [[-0.417574780492848, -0.175753709374146], [1.83958876341367, 1.8313809497999], [3, 5]],
                 [[-0.267522855927958, -0.161200683014406], [1.17559984533974, 1.21675019853961], [3, 5]])


def rew_consfactor(nw, n):
    if nw >= n or nw <= 0:
        return 1.0
    q = norm.ppf((nw + n) / (2.0 * n))
    return 1.0 / math.sqrt(1.0 - (2.0 * n) / (nw / q) * norm.pdf(q))


def lts_scale_tables(P, alpha):
    """(h, raw factor, rew factor table indexed by number of unit weights)."""
    h = lts_h(P, alpha)
    raw = raw_consfactor(h, P) * raw_corfactor(LTS_DIM, P, alpha)
    rew = np.ones(P + 1)
    cor = rew_corfactor(LTS_DIM, P, alpha)
    for nw in range(1, P):
        rew[nw] = rew_consfactor(nw, P) * cor
    return h, raw, rew


def lts_post_process(tau, xij, zraw, alpha):
    """post_process [R]: raw scale -> weights -> WLS refit -> reweighted scale -> final weights.
    -> z_final (2,nits), weights (P,nits) uint8, sigma_tau (nits)."""
    P, nits = tau.shape
    h, rawfac, rewtab = lts_scale_tables(P, alpha)
    zfin = np.full((2, nits), np.nan)
    weights = np.ones((P, nits), dtype=np.uint8)
    sig = np.full(nits, np.nan)
    for jj in range(nits):
        z0, z1 = zraw[0, jj], zraw[1, jj]
        if not (np.isfinite(z0) and np.isfinite(z1)):
            continue
        t = tau[:, jj]
        r = (t - xij[:, 0] * z0) - xij[:, 1] * z1
        order = np.argsort(np.abs(r), kind='stable')
        rank = np.empty(P, dtype=int); rank[order] = np.arange(P)
        ssq = 0.0
        for k in range(P):
            if rank[k] < h:
                ssq = ssq + r[k] * r[k]
        s0 = math.sqrt(ssq / h) * rawfac
        if abs(s0) < LTS_ZERO_SCALE:
            w = np.abs(r) < LTS_ZERO_SCALE
            zf0, zf1 = z0, z1
            rf = r
        else:
            w = np.abs(r / s0) <= LTS_QUANTILE
            zf0, zf1 = _fit_masked(xij, t, w)
            zf0 = float(zf0); zf1 = float(zf1)
            rf = (t - xij[:, 0] * zf0) - xij[:, 1] * zf1
            nw = int(np.sum(w))
            ssw = 0.0
            for k in range(P):
                if w[k]:
                    ssw = ssw + rf[k] * rf[k]
            scale = math.sqrt(ssw / (nw - 1)) * rewtab[nw] if nw > 1 else 0.0
            if scale > 0:
                w = np.abs(rf / scale) <= LTS_QUANTILE
            # scale == 0: exact fit on the kept pairs, keep the first-stage weights
        zfin[:, jj] = (zf0, zf1)
        weights[:, jj] = w.astype(np.uint8)
        nw = int(np.sum(w))
        acc = 0.0
        for k in range(P):
            if w[k]:
                acc = acc + t[k] * rf[k]
        if nw > LTS_DIM:
            with np.errstate(invalid='ignore'):
                sig[jj] = np.sqrt(acc / (nw - LTS_DIM))
    return zfin, weights, sig


def stdict_from_weights(weights, idx_pair, t, nchans):
    """array_from_weights + stdict packing [R]: key str(t), value 1-based element numbers of both
    members of every zero-weight pair (first members, then second members); 'size' = nchans."""
    stdict = {}
    idx = np.array(idx_pair)
    for jj in range(weights.shape[1]):
        drop = np.where(weights[:, jj] == 0)[0]
        if len(drop) > 0:
            stdict[str(t[jj])] = np.concatenate((idx[drop, 0] + 1, idx[drop, 1] + 1))
    stdict['size'] = nchans
    return stdict


def confidence_intervals(xij, z, sigma_tau, nphi=400000):
    """Szuberla & Olson (2004) 90 % confidence intervals [R: lts_array solve()/rthEllipse], restated from
    the geometric definition and evaluated by brute force: the confidence ellipse (semi-axes
    sqrt(chi2_0.90,2) sigma_tau / sqrt(eig(X^T X)) along the eigenvectors, centred on z) is sampled
    densely; conf_int_vel = half the spread of 1/|s|, conf_int_baz = half the angle it subtends at the
    origin (NaN if the origin is inside).  z: (2, nits)."""
    from scipy.stats import chi2
    nits = z.shape[1]
    civ = np.full(nits, np.nan)
    cib = np.full(nits, np.nan)
    evals, evecs = np.linalg.eigh(xij.T @ xij)
    ang = np.arccos(evecs[0, 0])
    R = np.array([[np.cos(ang), np.sin(ang)], [-np.sin(ang), np.cos(ang)]])
    c2 = chi2.ppf(0.90, 2)
    phi = np.linspace(0, 2 * np.pi, nphi, endpoint=False)
    for jj in range(nits):
        if not (np.isfinite(z[0, jj]) and np.isfinite(z[1, jj]) and np.isfinite(sigma_tau[jj])):
            continue
        a = np.sqrt(c2) * sigma_tau[jj] / np.sqrt(evals[0])
        b = np.sqrt(c2) * sigma_tau[jj] / np.sqrt(evals[1])
        so = R @ z[:, jj]
        if a == 0 and b == 0:
            civ[jj] = 0.0
            cib[jj] = 0.0
            continue
        pts = np.stack((so[0] + a * np.cos(phi), so[1] + b * np.sin(phi)), axis=1) @ R
        r = np.hypot(pts[:, 0], pts[:, 1])
        civ[jj] = 0.5 * abs(1.0 / r.min() - 1.0 / r.max())
        if (so[0] / a) ** 2 + (so[1] / b) ** 2 <= 1.0:
            continue                                   # origin inside the ellipse: direction undetermined
        cdir = z[:, jj] / np.hypot(*z[:, jj])
        rel = np.arctan2(cdir[0] * pts[:, 1] - cdir[1] * pts[:, 0], cdir[0] * pts[:, 0] + cdir[1] * pts[:, 1])
        cib[jj] = 0.5 * np.degrees(rel.max() - rel.min())
    return civ, cib


CHI2_90_2DOF = -2.0 * np.log(1.0 - 0.90)          # chi2.ppf(0.90, 2) = 4.605170185988092


def confidence_intervals_closed_form(xij, z, sigma_tau, nsample=720, newton_iters=8):
    """The same two quantities as ``confidence_intervals`` — which samples the ellipse densely and is the INDEPENDENT
    check — evaluated the way the GPU's ``uncertainty_kernel`` (csrc/solve.hip) evaluates them, operation for
    operation, so that the two can be compared to rounding: the radial extrema of the ellipse as stationary points of
    f(phi) = |c + (a cos phi, b sin phi)|^2 (best of ``nsample`` boundary samples, then ``newton_iters`` clipped Newton
    steps on f'(phi) = 0), the subtended angle from the two tangents through the origin in closed form (unit circle
    after scaling by the semi-axes).  [R: lts_array solve()/rthEllipse; Szuberla & Olson 2004.]  z: (2, nits)."""
    z = np.asarray(z, dtype=np.float64)
    sig = np.asarray(sigma_tau, dtype=np.float64).reshape(-1)
    n = len(sig)
    ci_vel = np.full(n, np.nan)
    ci_baz = np.full(n, np.nan)
    if n == 0:
        return ci_vel, ci_baz
    evals, evecs = np.linalg.eigh(xij.T @ xij)
    ang = np.arccos(np.clip(evecs[0, 0], -1.0, 1.0))
    R = np.array([[np.cos(ang), np.sin(ang)], [-np.sin(ang), np.cos(ang)]])
    q = np.sqrt(CHI2_90_2DOF)
    with np.errstate(invalid='ignore', divide='ignore', over='ignore'):
        a = q * sig / np.sqrt(evals[0])
        b = q * sig / np.sqrt(evals[1])
        x0 = z[0] * R[0, 0] + z[1] * R[0, 1]            # ellipse centre in the eigen-frame
        y0 = z[0] * R[1, 0] + z[1] * R[1, 1]
        ok = np.isfinite(a) & np.isfinite(b) & np.isfinite(x0) & np.isfinite(y0)
        step = 6.283185307179586 / nsample
        phi = np.arange(nsample) * step
        cx = x0[:, None] + a[:, None] * np.cos(phi)[None, :]
        cy = y0[:, None] + b[:, None] * np.sin(phi)[None, :]
        f = cx * cx + cy * cy
        f = np.where(np.isfinite(f), f, 0.0)
        r_ext = []
        for pick in (np.argmin, np.argmax):
            p = pick(f, axis=1) * step
            for _ in range(newton_iters):
                s, co = np.sin(p), np.cos(p)
                g = -a * s * (x0 + a * co) + b * co * (y0 + b * s)                     # f'/2
                c2 = co * co - s * s
                h = -a * co * x0 - a * a * c2 - b * s * y0 + b * b * c2                # f''/2
                st = np.where(np.abs(h) > 0, g / h, 0.0)
                p = p - np.clip(st, -0.05, 0.05)
            r_ext.append(np.hypot(x0 + a * np.cos(p), y0 + b * np.sin(p)))
        rmin, rmax = r_ext
        ci_vel = 0.5 * np.abs(1.0 / rmin - 1.0 / rmax)
        pz = np.where(a > 0, x0 / a, np.inf)
        qz = np.where(b > 0, y0 / b, np.inf)
        d2 = pz * pz + qz * qz
        outside = d2 > 1.0
        root = np.sqrt(np.where(outside, d2 - 1.0, np.nan)) / d2
        k = 1.0 - 1.0 / d2
        t1x, t1y = a * (pz * k - root * qz), b * (qz * k + root * pz)
        t2x, t2y = a * (pz * k + root * qz), b * (qz * k - root * pz)
        e1x, e1y = t1x * R[0, 0] + t1y * R[1, 0], t1x * R[0, 1] + t1y * R[1, 1]       # back to east / north: t @ R
        e2x, e2y = t2x * R[0, 0] + t2y * R[1, 0], t2x * R[0, 1] + t2y * R[1, 1]
        th1 = (np.arctan2(e1y, e1x) * (180.0 / 3.141592653589793) - 360.0) % 360.0
        th2 = (np.arctan2(e2y, e2x) * (180.0 / 3.141592653589793) - 360.0) % 360.0
        dth = np.abs(th1 - th2)
        dth = np.where(dth > 180.0, np.abs(dth - 360.0), dth)
        ci_baz = np.where(outside, 0.5 * dth, np.nan)
        zero = ok & (a == 0) & (b == 0)                 # exact fit: a point, no spread
        ci_vel = np.where(zero, 0.0, ci_vel)
        ci_baz = np.where(zero, 0.0, ci_baz)
    return np.where(ok, ci_vel, np.nan), np.where(ok, ci_baz, np.nan)


def ltsva(st, lat_list, lon_list, window_length, window_overlap, alpha=1.0,
          plot_array_coordinates=False, rij=None, return_internals=False, ci_samples=20000):
    """lts_array.ltsva [R] -> (vel, baz, t, mdccm, stdict, sigma_tau, conf_int_vel, conf_int_baz).

    ``rij`` (2,N) km overrides the lat/lon geometry (synthetic arrays); ``ci_samples`` is the boundary
    sampling of the brute-force confidence-interval evaluation."""
    nchans = len(st)
    if nchans < 3:
        raise RuntimeError('ltsva needs at least 3 array elements.')
    if alpha < 1.0 and nchans < 4:
        raise RuntimeError('LTS (alpha < 1) needs at least 4 array elements.')
    if not (0.5 <= alpha <= 1.0):
        raise ValueError('alpha must be in [0.5, 1.0]')
    npts = len(st[0].data)
    for tr in st:
        if len(tr.data) != npts:
            raise ValueError('All traces must have the same length.')
    fs = st[0].stats.sampling_rate
    if rij is None:
        rij = get_rij(lat_list, lon_list, nchans)
    W, inc, intervals = window_plan(npts, fs, window_length, window_overlap)
    tvec = times_matplotlib(st[0])
    data = np.empty((npts, nchans))
    for i, tr in enumerate(st):
        data[:, i] = tr.data
    xij, idx_pair = co_array(rij)
    if np.linalg.matrix_rank(xij) < LTS_DIM:
        raise RuntimeError('Co-array is ill posed for the least squares problem.')
    nits = len(intervals)
    t = np.array([tvec[t0 + int(W / 2)] for t0 in intervals]) if nits else np.zeros(0)
    tau, mdccm, cmax = correlate_windows(data, W, intervals, idx_pair, fs)
    if alpha == 1.0:
        z, vel, baz, sigma_tau = ols_solve(xij, tau)
        stdict = {}
        weights = np.ones((len(idx_pair), nits), dtype=np.uint8)
    else:
        zraw = fast_lts(tau, xij, alpha)
        z, weights, sigma_tau = lts_post_process(tau, xij, zraw, alpha)
        vel, baz = vel_baz(z)
        stdict = stdict_from_weights(weights, idx_pair, t, nchans)
    # ci_samples > 0: the brute-force evaluation (dense boundary sampling); 0: the stationary-point / tangent form the
    # GPU kernel follows (compared with each other in tests/test_host.py)
    if ci_samples:
        civ, cib = confidence_intervals(xij, z, sigma_tau, nphi=ci_samples)
    else:
        civ, cib = confidence_intervals_closed_form(xij, z, sigma_tau)
    out = (vel, baz, t, mdccm, stdict, sigma_tau, civ, cib)
    if return_internals:
        return out, dict(tau=tau, cmax=cmax, z=z, weights=weights, W=W, inc=inc,
                         intervals=intervals, xij=xij, idx_pair=idx_pair)
    return out


# --------------------------------------------------------------------------
# narrow_band_least_squares.py restatement
# --------------------------------------------------------------------------
def vector_len_of(WINLEN_list, WINOVER, st):
    """narrow_band_least_squares.py:41-47."""
    max_WINLEN = WINLEN_list[-1]
    sampinc = int((1 - WINOVER) * max_WINLEN)
    npts = len(st[0].data)
    its = np.arange(0, npts, sampinc)
    nits = len(its) - 1
    Fs = st[0].stats.sampling_rate
    return int(nits / Fs)


def narrow_band_least_squares(WINLEN_list, WINOVER, ALPHA, st, lat_list, lon_list, NBANDS, w, h,
                              freqlist, FREQ_BAND_TYPE, freq_resp_list, FILTER_TYPE, FILTER_ORDER,
                              FILTER_RIPPLE, rij=None):
    """narrow_band_least_squares.py:8-127 (padding rows are zeros as in the parallel variant,
    :268-272; the serial original leaves them uninitialised)."""
    vector_len = vector_len_of(WINLEN_list, WINOVER, st)
    vel_array = np.zeros((NBANDS, vector_len))
    baz_array = np.zeros((NBANDS, vector_len))
    mdccm_array = np.zeros((NBANDS, vector_len))
    sig_tau_array = np.zeros((NBANDS, vector_len))
    t_array = np.zeros((NBANDS, vector_len))
    stdict_all = {}
    w_array = np.zeros((NBANDS, len(w)), dtype=complex)
    h_array = np.zeros((NBANDS, len(h)), dtype=complex)
    num_compute_list = []
    for ii in range(NBANDS):
        if FREQ_BAND_TYPE == '2_octave_over':
            tempfmin, tempfmax = freqlist[ii], freqlist[ii + 2]
        else:
            tempfmin, tempfmax = freqlist[ii], freqlist[ii + 1]
        stf, Fs, sos = filter_data(st, FILTER_TYPE, tempfmin, tempfmax, FILTER_ORDER, FILTER_RIPPLE)
        ww, hh = signal.sosfreqz(sos, freq_resp_list, fs=Fs)
        w_array[ii, :] = ww
        h_array[ii, :] = hh
        temp_BT = WINLEN_list[ii] * (tempfmax - tempfmin)
        if temp_BT < 5.0:
            print('CAUTION: BT < 5! Band between ' + str(tempfmin) + ' Hz and ' + str(tempfmax)
                  + ' Hz has BT = ' + str(temp_BT))
        vel, baz, t, mdccm, stdict, sig_tau, _, _ = ltsva(stf, lat_list, lon_list, WINLEN_list[ii],
                                                          WINOVER, ALPHA, rij=rij)
        n = len(vel)
        vel_array[ii, :n] = make_float(vel)
        baz_array[ii, :n] = make_float(baz)
        mdccm_array[ii, :n] = make_float(mdccm)
        t_array[ii, :n] = make_float(t)
        num_compute_list.append(n)
        if ALPHA == 1.0:
            sig_tau_array[ii, :n] = make_float(sig_tau)
            stdict_all = None
        elif ALPHA < 1.0:
            temp = {}
            for key in stdict:
                if key != 'size':
                    temp[str(ii + 1).zfill(2) + '_' + key] = stdict[key]
                else:
                    temp[key] = stdict[key]
            stdict_all = {**stdict_all, **temp}
    return (vel_array, baz_array, mdccm_array, t_array, stdict_all, sig_tau_array,
            num_compute_list, w_array, h_array)


# --------------------------------------------------------------------------
# synthetic data — shared by tests and bench through tools/, kept here so the
# oracle is self-contained for the CPU baseline
# --------------------------------------------------------------------------
def brute_force_lts_objective(tau_col, xij, alpha):
    """Exact LTS by enumeration of every h-subset (P <= ~16).  Independent check of FAST-LTS:
    returns (best objective, z) in ORIGINAL (unstandardised) units."""
    import itertools
    P = len(tau_col)
    h = lts_h(P, alpha)
    best, bz = np.inf, None
    for sub in itertools.combinations(range(P), h):
        sub = list(sub)
        z, res, _, _ = np.linalg.lstsq(xij[sub], tau_col[sub], rcond=None)
        r = tau_col[sub] - xij[sub] @ z
        o = float(r @ r)
        if o < best:
            best, bz = o, z
    return best, bz
