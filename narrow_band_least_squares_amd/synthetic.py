"""Offline synthetic array data (stands in for ``waveform_collection.gather_waveforms``,
reference example.py:91, which needs the network).

A band-limited Gaussian-noise plane wave crossing an N-element array, plus independent
white noise per element; optionally one element gets a constant timing error so that its
N-1 pairs are LTS outliers.  Generator parameters follow SURVEY.md §8(d).
"""
import numpy as np

from .stream import Stream, Trace, Stats

SEED = 20220042


def array_geometry(nchans, radius_km=1.0, seed=SEED):
    """(2, N) element coordinates in km, element 0 at the centre, the rest uniform in a disc.
    NOT mean-removed (``get_rij`` output is; differences r_i - r_j are what matter)."""
    rng = np.random.default_rng(seed)
    r = radius_km * np.sqrt(rng.uniform(size=nchans))
    th = 2 * np.pi * rng.uniform(size=nchans)
    rij = np.vstack((r * np.cos(th), r * np.sin(th)))
    rij[:, 0] = 0.0
    return rij


def latlon_from_rij(rij, lat0=64.87, lon0=-147.86):
    """Small-offset conversion of km offsets to lat/lon lists (for the ``get_rij`` path)."""
    lat = lat0 + rij[1] / 111.2
    lon = lon0 + rij[0] / (111.32 * np.cos(np.radians(lat0)))
    return list(lat), list(lon)


def plane_wave(rij, npts, fs, fmin, fmax, baz_deg=225.0, vel_kms=0.34, snr_db=6.0,
               timing_error_s=0.0, bad_element=None, seed=SEED + 1, dtype=np.float64):
    """(N, npts) traces.  Element i records s(t - u.r_i / v) + noise, u = propagation direction
    (from back-azimuth ``baz_deg`` towards the array)."""
    rng = np.random.default_rng(seed)
    nchans = rij.shape[1]
    baz = np.radians(baz_deg)
    u = -np.array([np.sin(baz), np.cos(baz)])
    delays = (u @ rij) / vel_kms                      # seconds
    if bad_element is not None and timing_error_s != 0.0:
        delays = delays.copy()
        delays[bad_element] += timing_error_s
    nfft = npts
    freqs = np.fft.rfftfreq(nfft, d=1.0 / fs)
    spec = rng.standard_normal(len(freqs)) + 1j * rng.standard_normal(len(freqs))
    spec[(freqs < fmin) | (freqs > fmax)] = 0.0
    spec[0] = 0.0
    base = np.fft.irfft(spec, n=nfft)
    spec = spec / base.std()
    out = np.empty((nchans, npts), dtype=dtype)
    noise_amp = 10.0 ** (-snr_db / 20.0)
    for i in range(nchans):
        sig = np.fft.irfft(spec * np.exp(-2j * np.pi * freqs * delays[i]), n=nfft)
        out[i] = sig + noise_amp * rng.standard_normal(npts)
    return out


def make_stream(data, fs, starttime=17884.0729166667, lat=None, lon=None):
    """(N, npts) array -> duck-typed Stream (start time as a matplotlib date number;
    default 2018-12-19T01:45:00 as in example.py:46)."""
    st = Stream()
    for i in range(data.shape[0]):
        st.append(Trace(np.array(data[i], dtype=np.float64),
                        Stats(sampling_rate=fs, npts=data.shape[1], starttime=starttime,
                              latitude=None if lat is None else lat[i],
                              longitude=None if lon is None else lon[i])))
    return st


# The configurations of BASELINE.json / SURVEY.md §8(d)
CONFIGS = {
    # name: N, bands, band type, fmin, fmax, alpha, fs, duration_s, winlen_s, overlap, filter, order, ripple, radius
    'cfg1': dict(N=6, B=10, band_type='linear', fmin=0.5, fmax=5.0, alpha=1.0, fs=20.0, dur=1200.0,
                 winlen=30.0, overlap=0.5, ftype='butter', order=2, ripple=0.01, radius=1.0),
    'cfg1b': dict(N=8, B=8, band_type='log', fmin=0.1, fmax=5.0, alpha=1.0, fs=20.0, dur=1200.05,
                  winlen=(60, 30), overlap=0.5, ftype='cheby1', order=2, ripple=0.01, radius=1.0),
    'cfg2': dict(N=6, B=24, band_type='log', fmin=0.1, fmax=5.0, alpha=0.75, fs=20.0, dur=3600.0,
                 winlen=30.0, overlap=0.5, ftype='butter', order=2, ripple=0.01, radius=1.0),
    'cfg3': dict(N=8, B=48, band_type='log', fmin=0.1, fmax=10.0, alpha=0.5, fs=40.0, dur=21600.0,
                 winlen=30.0, overlap=0.5, ftype='butter', order=2, ripple=0.01, radius=1.0),
    'cfg4': dict(N=16, B=96, band_type='log', fmin=0.1, fmax=10.0, alpha=0.5, fs=100.0, dur=86400.0,
                 winlen=30.0, overlap=0.5, ftype='butter', order=2, ripple=0.01, radius=2.0),
    'cfg5': dict(N=32, B=128, band_type='log', fmin=0.1, fmax=5.0, alpha=0.5, fs=20.0, dur=3600.0,
                 winlen=30.0, overlap=0.5, ftype='butter', order=2, ripple=0.01, radius=2.0),
}


def build_config(name, scale=1.0, trace_seed=SEED + 1):
    """-> dict(st, rij, lat, lon, freqlist, WINLEN_list, ...) for a named configuration.
    ``scale`` < 1 shortens the trace (for tests)."""
    from .helpers import get_freqlist, get_winlenlist
    c = dict(CONFIGS[name])
    npts = int(round(c['dur'] * scale * c['fs']))
    rij = array_geometry(c['N'], c['radius'])
    lts = c['alpha'] < 1.0
    data = plane_wave(rij, npts, c['fs'], c['fmin'], c['fmax'],
                      timing_error_s=0.25 if lts else 0.0,
                      bad_element=c['N'] - 1 if lts else None, seed=trace_seed)
    lat, lon = latlon_from_rij(rij)
    st = make_stream(data, c['fs'], lat=lat, lon=lon)
    freqlist, nbands, fmax = get_freqlist(c['fmin'], c['fmax'], c['band_type'], c['B'])
    if isinstance(c['winlen'], tuple):
        winlens = get_winlenlist('adaptive', nbands, 50, c['winlen'][0], c['winlen'][1])
    else:
        winlens = get_winlenlist('constant', nbands, c['winlen'], c['winlen'], c['winlen'])
    c.update(st=st, rij=rij - rij.mean(axis=1, keepdims=True), lat=lat, lon=lon, freqlist=freqlist,
             NBANDS=nbands, WINLEN_list=winlens, data=data, npts=npts)
    return c
