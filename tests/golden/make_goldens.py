"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE in this container.

Run with the interpreter the reference's outer loop still works on (numpy < 2, because of
``dtype='complex_'`` at narrow_band_least_squares.py:58):

    /opt/conda/bin/python3.9 tests/golden/make_goldens.py loops
    /opt/conda/bin/python3.9 tests/golden/make_goldens.py extra     (round-2 additions only)
    /opt/conda/bin/python3.9 tests/golden/make_goldens.py 2octave   (round 4: the overlapping-band LTS fixture again)
    python tests/golden/make_goldens.py planners

What comes from the reference's own code: ``helpers.get_freqlist`` / ``get_winlenlist`` /
``make_float`` / ``get_rij`` (geometry arithmetic around the Vincenty call), ``filter_data``'s
cheby1 branch, and the whole band loop / padding / ``num_compute_list`` / ``stdict`` key
prefixing of ``narrow_band_least_squares`` and ``narrow_band_least_squares_parallel``.
What does NOT (absent third-party code): ``lts_array.ltsva`` and ``obspy`` are
stub modules backed by the CPU oracle, so these fixtures pin the PACKING against the reference and
the numerics against the oracle only.  The reference files are read in place from /root/reference
and never copied; only inputs and outputs are stored.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import nbls_oracle as o  # noqa: E402

# ---- stub modules for the absent dependencies ----
obspy = types.ModuleType('obspy')
geod = types.ModuleType('obspy.geodetics')
base = types.ModuleType('obspy.geodetics.base')
base.calc_vincenty_inverse = o.vincenty_inverse
obspy.geodetics = geod
geod.base = base
sys.modules.update({'obspy': obspy, 'obspy.geodetics': geod, 'obspy.geodetics.base': base})
lts = types.ModuleType('lts_array')
RIJ = {}


def _ltsva(st, lat, lon, winlen, winover, alpha, *a):
    return o.ltsva(st, lat, lon, winlen, winover, alpha, rij=RIJ.get('rij'))


lts.ltsva = _ltsva
sys.modules['lts_array'] = lts
sys.path.insert(0, '/root/reference')
import helpers as ref_helpers  # noqa: E402
import narrow_band_least_squares as ref  # noqa: E402


class RefStream(o.OStream):
    """obspy-Stream stand-in with the two methods helpers.filter_data calls (obspy recipe [R])."""

    def copy(self):
        return RefStream(o.OStream.copy(self))

    def filter(self, kind, freqmin, freqmax, corners, zerophase):
        from scipy import signal
        assert kind == 'bandpass' and zerophase
        fs = self[0].stats.sampling_rate
        sos, _, _ = o.design_bandpass('butter', freqmin, freqmax, corners, None, fs)
        for tr in self:
            first = signal.sosfilt(sos, tr.data)
            tr.data = signal.sosfilt(sos, first[::-1])[::-1]

    def taper(self, max_percentage):
        for tr in self:
            tr.data = tr.data * o.taper_window(len(tr.data), max_percentage)


def synth(nchans, npts, fs, fmin, fmax, bad=None, seed=7):
    rng = np.random.default_rng(seed)
    r = np.sqrt(rng.uniform(size=nchans)); th = 2 * np.pi * rng.uniform(size=nchans)
    rij = np.vstack((r * np.cos(th), r * np.sin(th))); rij[:, 0] = 0
    baz = np.radians(225.0); u = -np.array([np.sin(baz), np.cos(baz)])
    delays = (u @ rij) / 0.34
    if bad is not None:
        delays[bad] += 0.25
    freqs = np.fft.rfftfreq(npts, 1.0 / fs)
    spec = rng.standard_normal(len(freqs)) + 1j * rng.standard_normal(len(freqs))
    spec[(freqs < fmin) | (freqs > fmax)] = 0
    spec /= np.fft.irfft(spec, n=npts).std()
    data = np.array([np.fft.irfft(spec * np.exp(-2j * np.pi * freqs * d), n=npts) for d in delays])
    data += 0.5 * rng.standard_normal(data.shape)
    return data, rij - rij.mean(axis=1, keepdims=True)


def planners():
    out = {}
    for kind, args in (('linear', (0.5, 5.0, 10)), ('log', (0.1, 5.0, 8)), ('octave', (0.1, 5.0, 8)),
                       ('2_octave_over', (0.1, 5.0, 8)), ('onethird_octave', (0.5, 5.0, 8)),
                       ('octave_linear', (0.1, 5.0, 10))):
        fl, nb, fmax = ref_helpers.get_freqlist(args[0], args[1], kind, args[2])
        out['freqlist_' + kind] = dict(args=list(args), freqlist=[float(x) for x in fl], nbands=int(nb),
                                       fmax=float(fmax))
    out['winlen_adaptive'] = [int(x) for x in ref_helpers.get_winlenlist('adaptive', 8, 50, 60, 30)]
    out['winlen_constant'] = [int(x) for x in ref_helpers.get_winlenlist('constant', 5, 50, 60, 30)]
    lat = [64.87, 64.875, 64.868, 64.872]
    lon = [-147.86, -147.85, -147.87, -147.855]
    out['get_rij'] = dict(lat=lat, lon=lon, rij=ref_helpers.get_rij(lat, lon, 4).tolist())
    with open(os.path.join(HERE, 'planners.json'), 'w') as f:
        json.dump(out, f, indent=1)


def band_loop(name, nchans, npts, fs, fmin, fmax, nbands, band_type, ftype, winlens, alpha, bad=None):
    from joblib import parallel_backend
    data, rij = synth(nchans, npts, fs, fmin, fmax, bad=bad)
    RIJ['rij'] = rij
    st = RefStream(o.make_stream(data, fs, starttime=17884.0729166667))
    freqlist, nb, _ = ref_helpers.get_freqlist(fmin, fmax, band_type, nbands)
    if winlens == 'adaptive':
        wl = ref_helpers.get_winlenlist('adaptive', nb, 50, 60, 30)
    else:
        wl = ref_helpers.get_winlenlist('constant', nb, winlens, 0, 0)
    fr = np.logspace(-2, np.log10(fs / 2), 64)
    w = np.zeros(64)
    lat = [0.0] * nchans
    args = (wl, 0.5, alpha, st, lat, lat, nb, w, w, freqlist, band_type, fr, ftype, 2, 0.01)
    ser = ref.narrow_band_least_squares(*args)
    with parallel_backend('threading'):
        par = ref.narrow_band_least_squares_parallel(*args)
    ncl = list(ser[6])
    assert ncl == list(par[6])
    vl = ser[0].shape[1]
    # the serial reference leaves the row tails uninitialised: zero them for storage
    grids = []
    for i in (0, 1, 2, 3, 5):
        g = np.array(ser[i], dtype=float)
        for b in range(nb):
            g[b, ncl[b]:] = 0.0
            if not (i == 5 and alpha < 1.0):
                np.testing.assert_array_equal(g[b, :ncl[b]], par[i][b, :ncl[b]])
        grids.append(g)
    sd = ser[4]
    assert (sd is None) == (par[4] is None)
    keys, vals = [], []
    if sd is not None:
        assert set(sd.keys()) == set(par[4].keys())
        for k in sd:
            if k != 'size':
                keys.append(k); vals.append(np.asarray(sd[k], dtype=np.int64))
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'), data=data, rij=rij, fs=fs, freqlist=np.asarray(freqlist, dtype=float),
        winlens=np.asarray(wl), alpha=alpha, band_type=band_type, ftype=ftype, freq_resp=fr,
        vel=grids[0], baz=grids[1], mdccm=grids[2], t=grids[3], sig=grids[4] if alpha == 1.0 else np.zeros_like(grids[0]),
        num_compute=np.asarray(ncl), vector_len=vl, w_array=np.asarray(par[7]), h_array=np.asarray(par[8]),
        stdict_keys=np.asarray(keys), stdict_vals=np.asarray(np.concatenate(vals) if vals else np.zeros(0, dtype=np.int64)),
        stdict_lens=np.asarray([len(v) for v in vals], dtype=np.int64),
        stdict_size=-1 if sd is None else sd['size'])
    print(name, 'vector_len', vl, 'num_compute', ncl, 'stdict entries', len(keys))


def txtfile():
    """Bytes written by the reference's own write_txtfile (helpers.py:161-182) for the grids of the
    loop_ols_butter_linear golden, and the arrays its read_txtfile (helpers.py:185-235) returns for that
    file: the product's text I/O is compared with both."""
    import contextlib
    import io
    import tempfile
    g = np.load(os.path.join(HERE, 'loop_ols_butter_linear.npz'), allow_pickle=False)
    d = tempfile.mkdtemp() + '/'
    with contextlib.redirect_stdout(io.StringIO()):
        ref_helpers.write_txtfile(d, 'ref', g['vel'], g['baz'], g['mdccm'], g['t'], g['freqlist'], list(g['num_compute']))
    text = open(d + 'ref.txt', 'rb').read()
    open(os.path.join(HERE, 'txtfile_ref.txt'), 'wb').write(text)
    out = ref_helpers.read_txtfile(d, 'ref')
    names = ('vel', 'baz', 'mdccm', 't', 'freqlist', 'num_compute', 'nbands', 'fmin', 'fmax')
    # read_txtfile leaves the row tails uninitialised (np.empty): store them zeroed
    arrs = {}
    ncl = np.asarray(out[5])
    for n, v in zip(names, out):
        v = np.array(v, dtype=float)
        if v.ndim == 2:
            for b in range(v.shape[0]):
                v[b, ncl[b]:] = 0.0
        arrs[n] = v
    np.savez_compressed(os.path.join(HERE, 'txtfile_read.npz'), **arrs)
    print('txtfile', len(text), 'bytes;', 'read back', arrs['vel'].shape, ncl.tolist())


def filter_golden():
    """Output of the REFERENCE's own filter_data (helpers.py:108-141), 'cheby1' branch — SciPy iirfilter + causal
    sosfilt per trace + the whole-trace 1 % taper (the taper is the stand-in stream's, obspy recipe [R]) — on a small
    synthetic stream: pins the product's filter kernel to reference code directly."""
    data, _ = synth(5, 7000, 20.0, 0.2, 4.0, seed=21)
    st = RefStream(o.make_stream(data, 20.0, starttime=17884.0729166667))
    stf, fs, sos = ref_helpers.filter_data(st, 'cheby1', 0.4, 2.5, 3, 0.05)
    out = np.array([tr.data for tr in stf])
    assert not np.shares_memory(out, data) and np.array_equal(np.array([tr.data for tr in st]), data)   # input untouched
    np.savez_compressed(os.path.join(HERE, 'filter_cheby1_ref.npz'), data=data, fs=fs, sos=sos, filtered=out,
                        fmin=0.4, fmax=2.5, order=3, ripple=0.05)
    print('filter_cheby1_ref', out.shape, sos.shape)


def two_octave():
    """Overlapping bands ('2_octave_over': band ii spans freqlist[ii] .. freqlist[ii + 2], narrow_band_least_squares.py:69-71)
    under LTS.  Round 4: alpha = 0.5 instead of 0.75 — with 6 elements (15 pairs, h = 12 at 0.75) the five pairs of the
    mistimed element could not all be trimmed, no window dropped anything and the reference's key-prefix code (:114-124)
    was pinned on an EMPTY dictionary; at 0.5 (h = 9) the dictionary has an entry for most windows."""
    band_loop('loop_lts_2octave', 6, 6000, 20.0, 0.25, 4.0, 4, '2_octave_over', 'cheby1', 30, 0.5, bad=5)


def extra():
    """Fixtures added in round 2 (same interpreter as `loops`): more than 99 bands — the reference's
    str(band).zfill(2) prefix becomes three characters, '100_', '101_' (narrow_band_least_squares.py:120) —
    and the text-file fixtures."""
    band_loop('loop_lts_101bands', 5, 1600, 20.0, 0.5, 5.0, 101, 'linear', 'butter', 30, 0.5, bad=4)
    txtfile()
    filter_golden()


if __name__ == '__main__':
    # `planners`: run under the default interpreter (numpy 2.x: np.logspace differs from numpy 1.26 in
    # the last bit, and the planners do import there); `loops`: needs numpy < 2 (see the docstring).
    what = sys.argv[1:] or ['planners', 'loops']
    if 'planners' in what:
        planners()
    if 'extra' in what:
        extra()
    if 'filter' in what:
        filter_golden()
    if '2octave' in what:
        two_octave()
    if 'loops' not in what:
        sys.exit(0)
    band_loop('loop_ols_cheby1_adaptive', 8, 24001, 20.0, 0.1, 5.0, 8, 'log', 'cheby1', 'adaptive', 1.0)
    band_loop('loop_ols_butter_linear', 6, 6000, 20.0, 0.5, 5.0, 5, 'linear', 'butter', 30, 1.0)
    band_loop('loop_lts_butter_octave', 6, 6000, 20.0, 0.25, 4.0, 4, 'octave', 'butter', 30, 0.5, bad=5)
    two_octave()
    extra()

