"""Host-side mirror of the reference's ``helpers.py`` (same names, argument order, return
tuples and error behaviour).  ``filter_data`` runs on the GPU; the planners and text I/O are
plain NumPy as in the reference.
"""
import math

import numpy as np

from . import engine, planner


def get_freqlist(FMIN, FMAX, FREQ_BAND_TYPE, NBANDS):
    """Narrow frequency band limits -> (freqlist, nbands_calc, FMAX_calc).  Reference:
    helpers.py:8-79 (including its quirks: ``math.log(x, 10)`` for the log edges, the inclusive
    ``arange`` end for linear bands, the duplicated switch edge of 'octave_linear')."""
    freqrange = FMAX - FMIN
    if FREQ_BAND_TYPE == 'linear':
        step = freqrange / NBANDS
        freqlist = np.arange(FMIN, FMAX + step, step)
        return freqlist, NBANDS, FMAX
    if FREQ_BAND_TYPE == 'log':
        freqlist = np.logspace(math.log(FMIN, 10), math.log(FMAX, 10), num=NBANDS + 1)
        return freqlist, NBANDS, FMAX
    if FREQ_BAND_TYPE in ('octave', '2_octave_over'):
        freqlist = [FMIN, ]
        while 2 * freqlist[-1] <= FMAX:
            freqlist.append(2 * freqlist[-1])
        less = 1 if FREQ_BAND_TYPE == 'octave' else 2
        return freqlist, int(len(freqlist)) - less, freqlist[-1]
    if FREQ_BAND_TYPE == 'onethird_octave':
        ratio = 2 ** (1. / 3.)
        freqlist = [FMIN, ]
        while freqlist[-1] * ratio <= FMAX:
            freqlist.append(freqlist[-1] * ratio)
        return freqlist, int(len(freqlist)) - 1, freqlist[-1]
    if FREQ_BAND_TYPE == 'octave_linear':
        switch_freq = 2
        freqlist = [FMIN, ]
        while 2 * freqlist[-1] <= switch_freq:
            freqlist.append(2 * freqlist[-1])
        nlin = NBANDS - len(freqlist)
        step = (FMAX - freqlist[-1]) / nlin
        freqlist = freqlist + list(np.arange(freqlist[-1], FMAX + step, step))
        return freqlist, int(len(freqlist)) - 1, FMAX
    raise UnboundLocalError("unknown FREQ_BAND_TYPE %r" % (FREQ_BAND_TYPE,))


def get_winlenlist(WINDOW_LENGTH_TYPE, NBANDS, WINLEN, WINLEN_1, WINLEN_X):
    """Window length per band.  Reference: helpers.py:83-104."""
    if WINDOW_LENGTH_TYPE == 'constant':
        return [WINLEN for _ in range(NBANDS)]
    if WINDOW_LENGTH_TYPE == 'adaptive':
        return [int(item) for item in np.linspace(WINLEN_1, WINLEN_X, num=NBANDS)]
    raise UnboundLocalError("unknown WINDOW_LENGTH_TYPE %r" % (WINDOW_LENGTH_TYPE,))


def filter_data(st, FILTER_TYPE, FMIN, FMAX, FILTER_ORDER, FILTER_RIPPLE):
    """Band-pass and taper the data on the GPU -> (stf, Fs, sos).  Reference: helpers.py:108-141
    ('butter': zero-phase Butterworth as obspy applies it; 'cheby1': causal Chebyshev-I SOS;
    then a 1 % Hann taper of the whole trace).  ``st`` is not modified."""
    rows, fs, _ = engine.stream_rows(st)
    sos_apply, zero_phase, sos_ret = planner.design_bandpass(FILTER_TYPE, FMIN, FMAX, FILTER_ORDER,
                                                            FILTER_RIPPLE, fs)
    h = engine.get_handle()
    h.set_trace_rows(rows, fs)
    npts = len(rows[0])
    tl, tr = planner.taper_ramps(npts)
    # filter-only plan: one dummy window, no geometry needed (stage mask 1)
    h.plan(sos_apply[None, :, :], zero_phase, tl, tr, [2], [max(1, npts)], 1)
    h.execute(stages=1)
    h.sync()
    filt = h.fetch_filtered(0)
    from .stream import Stream, Trace
    if isinstance(st, Stream):
        # (the reference copies the stream and then replaces every trace's samples, helpers.py:124,137: the same
        #  result without copying 8 bytes per sample twice — each trace gets its row of the fetched block)
        stf = Stream(Trace(filt[ii], tr.stats.copy()) for ii, tr in enumerate(st))
    else:
        stf = st.copy()
        for ii in range(len(stf)):
            stf[ii].data = filt[ii]
    return stf, fs, sos_ret


def make_float(input):
    """1-D float64 array of the values of ``input``.  Reference: helpers.py:145-158."""
    return np.array([float(input[jj]) for jj in range(len(input))])


def write_txtfile(save_dir, fname, vel_array, baz_array, mdccm_array, t_array, freqlist, num_compute_list):
    """Tab-separated results file, one row per (band, window).  Reference: helpers.py:161-182."""
    with open(save_dir + fname + '.txt', 'w') as f:
        f.write('Fmin \t Fmax \t Time \t Trace_vel \t Backaz \t MdCCM \n')
        for ii in range(len(freqlist) - 1):
            print((num_compute_list[ii]))
            for jj in range(num_compute_list[ii]):
                f.write(str(freqlist[ii]) + '\t' + str(freqlist[ii + 1]) + '\t' + str(t_array[ii, jj]) + '\t'
                        + str(vel_array[ii, jj]) + '\t' + str(baz_array[ii, jj]) + '\t'
                        + str(mdccm_array[ii, jj]) + '\n')


def read_txtfile(save_dir, fname):
    """Inverse of ``write_txtfile``.  Reference: helpers.py:185-235 (bands are recovered from the
    unique Fmin values, so overlapping bands do not round-trip — as in the reference)."""
    temp_file = np.genfromtxt(save_dir + fname + '.txt', skip_header=1, dtype='float')
    fmin_list = temp_file[:, 0]
    fmax_temp = temp_file[-1, 1]
    unique_freq, idx = np.unique(fmin_list, return_index=True)
    freqlist = np.append(unique_freq, fmax_temp)
    idx = np.append(idx, len(fmin_list))
    num_compute_list = np.diff(idx)
    FMIN = fmin_list[0]
    FMAX = fmax_temp
    vector_len = len(fmin_list) - idx[-2]
    nbands = len(freqlist) - 1
    vel_array = np.empty((nbands, vector_len))
    baz_array = np.empty((nbands, vector_len))
    mdccm_array = np.empty((nbands, vector_len))
    t_array = np.empty((nbands, vector_len))
    for ii in range(nbands):
        a, b = idx[ii], idx[ii + 1]
        n = b - a
        vel_array[ii, :n] = temp_file[a:b, 3]
        baz_array[ii, :n] = temp_file[a:b, 4]
        mdccm_array[ii, :n] = temp_file[a:b, 5]
        t_array[ii, :n] = temp_file[a:b, 2]
    return vel_array, baz_array, mdccm_array, t_array, freqlist, num_compute_list, nbands, FMIN, FMAX


def vincenty_inverse(lat1, lon1, lat2, lon2):
    """Vincenty's inverse geodesic on WGS84 -> (distance m, azimuth 1->2 deg, azimuth 2->1 deg).
    Stands in for ``obspy.geodetics.base.calc_vincenty_inverse`` (helpers.py:4,271)."""
    a = 6378137.0
    f = 1.0 / 298.257223563
    b = (1.0 - f) * a
    if lat1 == lat2 and lon1 == lon2:
        return 0.0, 0.0, 0.0
    L = (math.radians(lon2 - lon1) + math.pi) % (2 * math.pi) - math.pi
    U1 = math.atan((1 - f) * math.tan(math.radians(lat1)))
    U2 = math.atan((1 - f) * math.tan(math.radians(lat2)))
    sU1, cU1, sU2, cU2 = math.sin(U1), math.cos(U1), math.sin(U2), math.cos(U2)
    lam = L
    for _ in range(200):
        sl, cl = math.sin(lam), math.cos(lam)
        sin_sig = math.hypot(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl)
        if sin_sig == 0.0:
            return 0.0, 0.0, 0.0
        cos_sig = sU1 * sU2 + cU1 * cU2 * cl
        sig = math.atan2(sin_sig, cos_sig)
        sin_al = cU1 * cU2 * sl / sin_sig
        cos2_al = 1.0 - sin_al * sin_al
        cos2sm = cos_sig - 2.0 * sU1 * sU2 / cos2_al if cos2_al != 0.0 else 0.0
        Cc = f / 16.0 * cos2_al * (4.0 + f * (4.0 - 3.0 * cos2_al))
        new = L + (1.0 - Cc) * f * sin_al * (
            sig + Cc * sin_sig * (cos2sm + Cc * cos_sig * (-1.0 + 2.0 * cos2sm * cos2sm)))
        converged = abs(new - lam) < 1e-12
        lam = new
        if converged:
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
    """Zero-mean local (x east, y north) element coordinates in km from lat/lon via the Vincenty
    inverse from element 0 -> array (2, nchans).  Reference: helpers.py:239-284."""
    if (len(latlist) != nchans) or (len(lonlist) != nchans):
        raise ValueError('Mismatch between the number of stream channels and the latitude or longitude list length.')  # noqa
    xnew = np.zeros((nchans, ))
    ynew = np.zeros((nchans, ))
    for jj in range(1, nchans):
        delta, az, _ = vincenty_inverse(latlist[0], lonlist[0], latlist[jj], lonlist[jj])
        az = (450 - az) % 360
        xnew[jj] = delta / 1000 * np.cos(az * np.pi / 180)
        ynew[jj] = delta / 1000 * np.sin(az * np.pi / 180)
    xnew -= np.mean(xnew)
    ynew -= np.mean(ynew)
    return np.array([xnew.tolist(), ynew.tolist()])
