"""GPU parity tests at the unit SHAPES of every BASELINE.json configuration, through the whole drop-in call
(narrow_band_least_squares -> C ABI) against the CPU oracle: cfg-2 at its own alpha with all 24 bands, cfg-4
(16 elements, 100 Hz, W = 3000, alpha 0.5, 500 LCG starts), cfg-5 (32 elements, 128 bands: three-character
'100_' key prefixes).  Trace lengths are cut so that the oracle finishes in seconds; the arithmetic per unit is
the configuration's.  Tolerances as in test_gpu_parity.py: lags / weights / stdict exact, vel / baz 1e-9."""
import numpy as np
import pytest

from narrow_band_least_squares_amd import engine, narrow_band_least_squares, narrow_band_loop, synthetic
from narrow_band_least_squares_amd.narrow_band_least_squares import _band_prefix

from test_gpu_parity import _compare_nbls, _compare_ltsva, _cfg, _ostream, RTOL

pytestmark = pytest.mark.gpu


def _sub_config(c, bands):
    """The configuration restricted to a contiguous band range (same trace, same windows)."""
    d = dict(c)
    d['NBANDS'] = len(bands)
    d['freqlist'] = list(c['freqlist'][bands[0]:bands[-1] + 2])
    d['WINLEN_list'] = list(c['WINLEN_list'][bands[0]:bands[-1] + 1])
    return d


def _rows_vs_oracle(oracle, c, got, bands):
    """Rows and stdict entries of selected bands of a whole call against filter_data + ltsva of the oracle."""
    for b in bands:
        stf, _, _ = oracle.filter_data(_ostream(oracle, c), c['ftype'], c['freqlist'][b], c['freqlist'][b + 1], c['order'],
                                       c['ripple'])
        out = oracle.ltsva(stf, None, None, c['WINLEN_list'][b], c['overlap'], c['alpha'], rij=c['rij'])
        n = len(out[0])
        assert got[6][b] == n
        np.testing.assert_allclose(got[0][b, :n], out[0], rtol=RTOL, err_msg='vel band %d' % b)
        np.testing.assert_allclose(got[1][b, :n], out[1], rtol=RTOL, err_msg='baz band %d' % b)
        np.testing.assert_allclose(got[2][b, :n], out[3], rtol=RTOL, err_msg='mdccm band %d' % b)
        np.testing.assert_array_equal(got[3][b, :n], out[2])
        assert not got[0][b, n:].any() and not got[3][b, n:].any()
        pre = _band_prefix(b + 1)
        mine = {k: v for k, v in got[4].items() if k.startswith(pre) and k != 'size'}
        theirs = {pre + k: v for k, v in out[4].items() if k != 'size'}
        assert mine.keys() == theirs.keys(), 'stdict keys of band %d' % b
        for k in theirs:
            np.testing.assert_array_equal(mine[k], theirs[k])


def test_cfg2_all_bands_at_its_own_alpha(oracle):
    """cfg-2: 6 elements, 24 log bands 0.1-5 Hz, alpha = 0.75 (h = 12, all 105 two-subsets), 20 Hz, 30 s windows."""
    c = _cfg('cfg2', 0.2)
    assert c['alpha'] == 0.75 and c['NBANDS'] == 24
    got, exp = _compare_nbls(oracle, c, np.logspace(-2, 1, 64))
    assert sum(got[6]) == 24 * 46 and any(k != 'size' for k in got[4])


def test_cfg4_unit_shape_through_the_whole_call(oracle):
    """cfg-4's unit: 16 elements (P = 120, h = 61), 100 Hz, W = 3000 samples, alpha = 0.5, 500 LCG-random starts —
    3 bands x 21 windows through narrow_band_least_squares() against the oracle; lags and weights of one band
    exactly; the int8 screening correlator and the large-array LTS kernel are the ones that run."""
    c = _cfg('cfg4', 345.0 / 86400.0)
    assert c['N'] == 16 and c['fs'] == 100.0 and c['alpha'] == 0.5
    sub = _sub_config(c, [50, 51, 52])
    h = engine.get_handle()
    h.set_profiling(True)
    try:
        got, exp = _compare_nbls(oracle, sub, np.logspace(-2, np.log10(50.0), 64))
        assert h.timings()['xcorr_impl'] == 3
    finally:
        h.set_profiling(False)
    assert got[6] == [21, 21, 21]
    from narrow_band_least_squares_amd import planner
    lp = planner.lts_plan(planner.co_array(c['rij'])[0], 0.5)
    assert lp['h'] == 61 and lp['starts'].shape[0] == 500
    # lags and weights of the middle band, exactly
    stf, _, _ = oracle.filter_data(_ostream(oracle, c), 'butter', c['freqlist'][51], c['freqlist'][52], 2, 0.01)
    _compare_ltsva(oracle, c, stf, 30.0, 0.5)
    # the mistimed element shows up in the dropped-element dictionary
    dropped = np.concatenate([v for k, v in got[4].items() if k != 'size'])
    assert np.bincount(dropped, minlength=17).argmax() == 16


def test_cfg5_more_than_99_bands(oracle):
    """cfg-5: 32 elements (P = 496, h = 249, 500 LCG starts), 128 bands in ONE call: bands 100..128 get the
    three-character key prefix the reference's str(band).zfill(2) produces ('100_', narrow_band_least_squares.py:120).
    Selected bands (first, 99th, 100th, 101st, last) against the oracle; every band's keys carry its own prefix."""
    c = _cfg('cfg5', 75.0 / 3600.0)
    assert c['N'] == 32 and c['NBANDS'] == 128
    fr = np.logspace(-2, 1, 16)
    w = np.zeros(16)
    got = narrow_band_least_squares(c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, 128, w, w,
                                    c['freqlist'], c['band_type'], fr, c['ftype'], c['order'], c['ripple'], rij=c['rij'])
    assert got[6] == [3] * 128 and got[0].shape[0] == 128
    _rows_vs_oracle(oracle, c, got, [0, 98, 99, 100, 127])
    keys = [k for k in got[4] if k != 'size']
    assert any(k.startswith('100_') for k in keys) and any(k.startswith('128_') for k in keys)
    assert all(k.split('_')[0] == str(int(k.split('_')[0])).zfill(2) for k in keys)
    assert got[4]['size'] == 32


def test_narrow_band_loop_lts_branch(oracle):
    """narrow_band_loop() under LTS (reference narrow_band_least_squares.py:204-214): the dictionary of one
    band comes back flattened into two object arrays (times, elements) — the joblib transport form."""
    c = _cfg('cfg2', 0.2)
    fr = np.logspace(-2, 1, 32)
    vl = 60
    row = narrow_band_loop(7, c['freqlist'], c['band_type'], fr, c['st'], 'butter', 2, 0.01, None, None, c['WINLEN_list'], 0.5,
                           0.75, vl, rij=c['rij'])
    stf, _, _ = oracle.filter_data(_ostream(oracle, c), 'butter', c['freqlist'][7], c['freqlist'][8], 2, 0.01)
    out = oracle.ltsva(stf, None, None, c['WINLEN_list'][7], 0.5, 0.75, rij=c['rij'])
    n = len(out[0])
    assert int(row[7]) == n and len(row[0]) == vl
    np.testing.assert_allclose(row[0][:n], out[0], rtol=RTOL)
    np.testing.assert_allclose(row[1][:n], out[1], rtol=RTOL)
    np.testing.assert_array_equal(row[3][:n], out[2])
    times, elements = row[4], row[5]
    assert times.dtype == object and elements.dtype == object and len(times) == len(out[4])
    exp_items = list(out[4].items())
    assert list(times) == [k for k, _ in exp_items]
    for (k, v), e in zip(exp_items, elements):
        if k == 'size':
            assert e == 6
        else:
            np.testing.assert_array_equal(e, v)


def _hostile_stream(oracle, kind):
    """A pre-filtered 6-element (kind 'long': 4-element) stream that stresses the screening correlator."""
    rng = np.random.default_rng(11)
    fs, nchans, winlen = 20.0, 6, 30.0
    if kind == 'long':
        fs, nchans, winlen = 200.0, 4, 25.0                    # W = 5000 samples: the LDS-slab quantiser
    rij = synthetic.array_geometry(nchans, 1.0, seed=40 + nchans)
    npts = int(5.2 * winlen * fs)
    t = np.arange(npts) / fs
    if kind == 'noise':                                         # no common signal at all
        data = rng.standard_normal((nchans, npts))
    elif kind == 'sinusoid':                                    # periodic: every period is a near-tie
        delays = rng.uniform(0, 0.3, nchans)
        data = np.array([np.sin(2 * np.pi * 0.537 * (t - d)) for d in delays])
    else:
        data = synthetic.plane_wave(rij, npts, fs, 0.5, 0.4 * fs if kind != 'long' else 8.0, seed=5)
    W = int(winlen * fs)
    if kind == 'spike':                                         # one sample 1e6 x the signal in every window
        data[2, W // 3::W // 2] = 1e6
    elif kind == 'dc':
        data[4] += 250.0
    elif kind == 'ties':                                        # a constant against an alternating +-1 channel:
        data[0] = 1.0                                           # their correlation is exactly 0 or +-1 at every
        data[1] = (-1.0) ** np.arange(npts)                     # lag — hundreds of exactly tied maxima
    elif kind == 'nan':
        data[1, W + 17] = np.nan                                # windows 1 and 2 of channel 1
        data[3, 3 * W // 2 + 5] = np.nan
    elif kind == 'inf':
        data[1, W + 17] = np.inf
        data[5, W + 400] = -np.inf
        data[0, 3 * W + 3] = np.inf                             # last windows: one channel only
    c = dict(fs=fs, rij=rij - rij.mean(axis=1, keepdims=True))
    return c, oracle.make_stream(data, fs, starttime=17884.0729166667), winlen


@pytest.mark.parametrize('kind', ['noise', 'sinusoid', 'spike', 'dc', 'ties', 'long', 'nan', 'inf'])
@pytest.mark.parametrize('alpha', [1.0, 0.5])
def test_hostile_inputs_keep_exact_lags(oracle, kind, alpha):
    """Inputs that break the assumptions the int8 screening is tuned for — incoherent noise, a pure sinusoid
    (periodic near-ties), a 1e6 x spike per window (the 15-bit quantisation of everything else collapses to 0),
    a DC offset, exactly tied maxima (candidate overflow -> full-lag fallback), W > 4096 (slab quantiser) — and NaN / Inf samples, where the answer is NumPy's (first NaN lag
    wins, wave_ops.h nonfinite_argmax): lags, maxima, weights exactly the oracle's in every case."""
    c, st, winlen = _hostile_stream(oracle, kind)
    h = engine.get_handle()
    h.set_profiling(True)
    try:
        _compare_ltsva(oracle, c, st, winlen, alpha)
        assert h.timings()['xcorr_impl'] == 3                   # the screening correlator ran, not a fallback kernel
        stats = h.screen_stats()
    finally:
        h.set_profiling(False)
    assert stats['pairs'] > 0
    if kind == 'ties':
        # hundreds of lags tie for the maximum: the candidate lists of that pair overflow, so these windows really
        # went through the interval / full-lag FP64 fallback (and np.argmax's first-index rule decided)
        assert stats['overflow'] > 0, stats


def test_raw_trace_with_a_gap_propagates_like_sosfilt(oracle):
    """A NaN in the RAW trace: the IIR band-pass spreads it over the rest of that channel (forward pass) and,
    zero-phase, over all of it; every pair of that element then follows the NaN rule.  Same dropped pairs and
    same rows as the oracle (scipy.sosfilt + NumPy)."""
    c = _cfg('cfg1', 0.25)
    data = np.array(c['data'])
    data[2, 1234] = np.nan
    st = synthetic.make_stream(data, c['fs'], starttime=c['st'][0].stats.starttime)
    d = dict(c)
    d['st'], d['data'] = st, data
    sub = _sub_config(d, [2, 3])
    got, exp = _compare_nbls(oracle, sub, np.logspace(-2, 1, 16))
    assert np.isfinite(got[0][:, :got[6][0]]).all()            # the other five elements still give a solution


def test_long_windows_of_example_py_at_100_hz(oracle):
    """example.py's adaptive WINLEN_1 = 60 s on 100 Hz data (cfg-4's rate) is W = 6000 samples.  With 8 elements the
    int8 images of all seven partners no longer fit a CU's LDS next to the sliding channel's copies: the screening
    kernel now splits the partners into two groups of four (the mechanism that serves 18..32 elements) instead of
    handing the plan to the general FP64 correlator — lags, maxima and the solution exactly the oracle's."""
    fs, nchans, winlen = 100.0, 8, 60.0
    rij = synthetic.array_geometry(nchans, 1.0, seed=77)
    data = synthetic.plane_wave(rij, int(2.6 * winlen * fs), fs, 0.2, 4.0, seed=9)
    c = dict(fs=fs, rij=rij - rij.mean(axis=1, keepdims=True))
    st = oracle.make_stream(data, fs, starttime=17884.0729166667)
    h = engine.get_handle()
    h.set_profiling(True)
    try:
        _compare_ltsva(oracle, c, st, winlen, 1.0)
        tm = h.timings()
        assert tm['xcorr_impl'] == 3 and tm['xcorr_fallback_bands'] == 0
    finally:
        h.set_profiling(False)


@pytest.mark.parametrize('nchans,fs,winlen,noise,screened', [(8, 200.0, 60.0, False, True), (8, 200.0, 60.0, True, True),
                                                             (16, 150.0, 60.0, True, True), (5, 200.0, 75.0, True, False)])
def test_windows_beyond_ten_thousand_samples(oracle, nchans, fs, winlen, noise, screened):
    """VERDICT r03 "missing 4": the reference has no window limit (helpers.py:99-102; example.py:62 WINLEN_1 = 60 s at 200 Hz
    is 12 000 samples) — the library refused more than 10 000 and ran 7 900..10 000 on a 60x slower correlator.  Now the
    screening kernel keeps FOUR byte-shifted copies of the sliding channel where eight do not fit (half the LDS per sample:
    up to ~13 000 samples), and beyond that the general correlator reads its windows from global memory.  8 el. x 12 000
    and 16 el. x 9 000 samples, plane wave and incoherent noise, two windows each: lags, maxima, solution against the oracle;
    5 el. x 15 000 on the general correlator."""
    W = int(winlen * fs)
    rng = np.random.default_rng(W + nchans)
    rij = synthetic.array_geometry(nchans, 1.0, seed=60 + nchans)
    npts = int(2.1 * W)
    data = rng.standard_normal((nchans, npts)) if noise else synthetic.plane_wave(rij, npts, fs, 0.2, 8.0, seed=5)
    c = dict(fs=fs, rij=rij - rij.mean(axis=1, keepdims=True))
    st = oracle.make_stream(data, fs, starttime=17884.0729166667)
    h = engine.get_handle()
    h.set_profiling(True)
    try:
        out = _compare_ltsva(oracle, c, st, winlen, 1.0)
        tm = h.timings()
        assert len(out[0]) >= 2
        assert (tm['xcorr_impl'] == 3 and tm['xcorr_fallback_bands'] == 0) if screened else tm['xcorr_impl'] in (1, 2)
    finally:
        h.set_profiling(False)


@pytest.mark.parametrize('nchans,W,noise', [(8, 6000, False), (8, 6000, True), (16, 3200, True), (16, 4500, False), (16, 4500, True),
                                            (12, 5003, True), (5, 7400, True), (24, 3100, True),
                                            (8, 9000, True), (8, 12000, False), (16, 9000, False), (4, 12700, True), (24, 7003, True)])
def test_partner_groups_extend_the_screening_correlator_to_long_windows(nchans, W, noise):
    """Window lengths whose int8 images do not fit a CU's LDS with all partners in one workgroup: partner groups of
    8 / 4 / 2 (tile geometry: 2 / 4 / 8 lag blocks per tile column, linear skew).  Lags must be those of the plain
    VALU correlator, on a plane wave and on incoherent noise (arg-max anywhere among the 2W-1 lags: every lag block
    of every column is exercised), for window lengths that are not multiples of the tile steps too."""
    fs = 100.0
    rng = np.random.default_rng(W + nchans)
    rij = synthetic.array_geometry(nchans, 1.0, seed=nchans)
    npts = int(2.3 * W)
    data = rng.standard_normal((nchans, npts)) if noise else synthetic.plane_wave(rij, npts, fs, 0.5, 20.0, seed=3)
    kw = dict(want_lag=True, want_cmax=True)
    edges = [(0.5, 20.0)]
    wl = [W / fs + 1e-9]
    h = engine.get_handle()
    h.set_profiling(True)
    try:
        got = engine.process(data, fs, 0.0, rij, edges, wl, 0.5, 1.0, 'butter', 2, 0.01, **kw)
        tm = h.timings()
        assert tm['xcorr_impl'] == 3 and tm['xcorr_fallback_bands'] == 0
    finally:
        h.set_profiling(False)
    assert int(got.W[0]) == W
    ref = engine.process(data, fs, 0.0, rij, edges, wl, 0.5, 1.0, 'butter', 2, 0.01, xcorr_impl=1, **kw)
    if noise:
        assert np.abs(ref.lag).max() > W // 4
    np.testing.assert_array_equal(got.lag, ref.lag)
    np.testing.assert_allclose(got.cmax, ref.cmax, rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(got.baz, ref.baz)


def test_adaptive_windows_choose_the_correlator_per_window_length(oracle, monkeypatch):
    """example.py's adaptive windows 60 -> 30 s at 100 Hz in ONE call, plus a 140 s band (W = 14 000: beyond what the
    screening kernel can hold even with four copies and two partners per group, ~13 000): every band of a screenable
    window length is screened, only the 140 s band runs on the general correlator (windows read from global memory:
    they do not fit a CU's LDS either) — not the whole plan.  Rows against the oracle."""
    from narrow_band_least_squares_amd import helpers
    fs, nchans = 100.0, 8
    rij = synthetic.array_geometry(nchans, 1.0, seed=5)
    data = synthetic.plane_wave(rij, 40000, fs, 0.1, 5.0, seed=21)
    winlens = [140.0] + helpers.get_winlenlist('adaptive', 4, 50, 60, 30)
    edges = [(0.1, 0.3), (0.3, 0.6), (0.6, 1.2), (1.2, 2.4), (2.4, 4.8)]
    h = engine.get_handle()
    h.set_profiling(True)
    try:
        got = engine.process(data, fs, 17884.0729166667, rij - rij.mean(axis=1, keepdims=True), edges, winlens, 0.5, 1.0, 'butter', 2, 0.01,
                             want_lag=True, groups=1)          # ONE plan with all five bands, whatever NBLS_PIPELINE_GROUPS says
        tm = h.timings()
        assert tm['xcorr_impl'] == 3 and tm['xcorr_fallback_bands'] == 1
    finally:
        h.set_profiling(False)
    st = oracle.make_stream(data, fs, starttime=17884.0729166667)
    for b, (lo, hi) in enumerate(edges):
        stf, _, _ = oracle.filter_data(st, 'butter', lo, hi, 2, 0.01)
        out, internals = oracle.ltsva(stf, None, None, winlens[b], 0.5, 1.0, rij=rij - rij.mean(axis=1, keepdims=True), return_internals=True)
        n = int(got.nwin[b])
        assert n == len(out[0]) and n >= 1
        np.testing.assert_array_equal(got.lag[b, :n], np.rint(internals['tau'].T * fs).astype(int))
        np.testing.assert_allclose(got.vel[b, :n], out[0], rtol=1e-9)
        np.testing.assert_allclose(got.baz[b, :n], out[1], rtol=1e-9)
        np.testing.assert_allclose(got.mdccm[b, :n], out[3], rtol=1e-9)
    # with the solves per unit batch (option "overlap", and every streamed pass) the band on the general
    # correlator was never solved — all-zero rows with rc OK.  Same rows now, whichever way the solves are scheduled.
    ref_rows = {k: getattr(got, k).copy() for k in ('vel', 'baz', 'mdccm', 'sigma_tau', 'lag')}
    r0 = rij - rij.mean(axis=1, keepdims=True)
    monkeypatch.setenv('NBLS_STREAM_RESULTS', '1')      # (calls without `groups` below are streamed passes; by itself a call of this size is not)
    for alpha in (1.0, 0.5):
        base = engine.process(data, fs, 17884.0729166667, r0, edges, winlens, 0.5, alpha, 'butter', 2, 0.01, want_lag=True, want_z=True, groups=1)
        if alpha == 1.0:
            for k, v in ref_rows.items():
                np.testing.assert_array_equal(getattr(base, k), v)
        assert np.all(base.vel[0, :int(base.nwin[0])] > 0)
        try:
            h.set_option('overlap', 1)
            ov = engine.process(data, fs, 17884.0729166667, r0, edges, winlens, 0.5, alpha, 'butter', 2, 0.01, want_lag=True, want_z=True, groups=1)
            st_ov = engine.process(data, fs, 17884.0729166667, r0, edges, winlens, 0.5, alpha, 'butter', 2, 0.01, want_lag=True, want_z=True)
            h.set_option('overlap', -1)
            st_one = engine.process(data, fs, 17884.0729166667, r0, edges, winlens, 0.5, alpha, 'butter', 2, 0.01, want_lag=True, want_z=True)
        finally:
            h.set_option('overlap', 0)
        st = engine.process(data, fs, 17884.0729166667, r0, edges, winlens, 0.5, alpha, 'butter', 2, 0.01, want_lag=True, want_z=True)
        for other in (ov, st_ov, st_one, st):
            for k in ('vel', 'baz', 'mdccm', 'sigma_tau', 'lag', 'z', 'mask'):
                np.testing.assert_array_equal(getattr(other, k), getattr(base, k), err_msg=k)


@pytest.mark.parametrize('ftype,alpha', [('butter', 1.0), ('cheby1', 0.5)])
def test_time_segmented_path_when_one_band_exceeds_the_hbm_budget(oracle, monkeypatch, ftype, alpha):
    """with a filtered-trace budget smaller than ONE band the call runs band by band, the trace goes
    through the filter in time segments with the IIR state handed from segment to segment (forward, and backward
    for the zero-phase Butterworth), the windows in slices.  Same tuple as the in-core call (lags and dropped
    elements exactly, values to the rounding of the carried filter states) and as the oracle."""
    c = _cfg('cfg2', 0.15)                      # 6 elements, 10 800 samples
    c['alpha'], c['ftype'] = alpha, ftype
    sub = _sub_config(c, [10, 11, 12])
    fr = np.logspace(-2, 1, 32)
    w = np.zeros(32)
    args = (sub['WINLEN_list'], 0.5, alpha, c['st'], None, None, 3, w, w, sub['freqlist'], c['band_type'], fr, ftype, 2, 0.01)
    incore = narrow_band_least_squares(*args, rij=c['rij'])
    band_bytes = 8.0 * 6 * (c['npts'] + 64)
    monkeypatch.setenv('NBLS_MAX_FILTERED_GB', repr(0.45 * band_bytes / 2.0 ** 30))
    assert engine.max_bands_per_pass(6, c['npts']) == 0
    calls = {'n': 0}
    real = engine.filter_band_segmented

    def counting(h, rows, fs, sos, zp, seg_len):
        calls['n'] += 1
        assert seg_len < c['npts'] / 2 and seg_len % 1 == 0           # several segments per band
        return real(h, rows, fs, sos, zp, seg_len)
    monkeypatch.setattr(engine, 'filter_band_segmented', counting)
    seg = narrow_band_least_squares(*args, rij=c['rij'])
    assert calls['n'] == 3
    assert seg[6] == incore[6]
    for i in (0, 1, 2):
        np.testing.assert_allclose(seg[i], incore[i], rtol=1e-9, atol=0)
    np.testing.assert_array_equal(seg[3], incore[3])
    np.testing.assert_allclose(seg[5], incore[5], rtol=1e-7, atol=1e-12)
    np.testing.assert_array_equal(seg[7], incore[7])
    np.testing.assert_array_equal(seg[8], incore[8])
    if alpha < 1.0:
        assert list(seg[4].keys()) == list(incore[4].keys())
        for k in incore[4]:
            np.testing.assert_array_equal(seg[4][k], incore[4][k])
    # and against the oracle, through the same helper as every other whole-call test
    d = dict(sub)
    d['alpha'], d['ftype'] = alpha, ftype
    _compare_nbls(oracle, d, fr)


def test_filter_data_against_the_references_own_output():
    """tests/golden/filter_cheby1_ref.npz holds what the REFERENCE's filter_data (helpers.py:108-141, cheby1 branch)
    returned for a small stream: same SOS bit for bit, filtered traces to 1e-11 of the trace scale."""
    import os
    from narrow_band_least_squares_amd import filter_data
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'filter_cheby1_ref.npz'), allow_pickle=False)
    st = synthetic.make_stream(g['data'], float(g['fs']))
    stf, fs, sos = filter_data(st, 'cheby1', float(g['fmin']), float(g['fmax']), int(g['order']), float(g['ripple']))
    assert fs == float(g['fs'])
    np.testing.assert_array_equal(sos, g['sos'])
    out = np.array([tr.data for tr in stf])
    assert np.max(np.abs(out - g['filtered'])) <= 1e-11 * np.max(np.abs(g['filtered']))
    np.testing.assert_array_equal(np.array([tr.data for tr in st]), g['data'])         # the input stream is not modified


def test_result_block_stays_contiguous_across_plans(monkeypatch):
    """A smaller plan after a bigger one keeps the allocation but must move the grid views: vel | baz | mdccm |
    sigma_tau are always back to back (one D2H copy / one RCCL block), and the values are those of a fresh handle."""
    monkeypatch.setenv('NBLS_PIPELINE_GROUPS', '1')         # the plans under test are those of handle 0 with ALL bands
    c = _cfg('cfg1', 0.3)
    fr = np.logspace(-2, 1, 16)
    w = np.zeros(16)

    def call(nb):
        return narrow_band_least_squares(c['WINLEN_list'][:nb], 0.5, 1.0, c['st'], None, None, nb, w, w, c['freqlist'][:nb + 1],
                                         c['band_type'], fr, 'butter', 2, 0.01, rij=c['rij'])
    big = call(10)
    h = engine.get_handle()
    ptrs, nbytes = h.device_results()
    assert all(ptrs[i + 1] - ptrs[i] == nbytes for i in range(3))
    small = call(3)
    ptrs, nbytes = h.device_results()
    assert nbytes == 3 * small[0].shape[1] * 8 and all(ptrs[i + 1] - ptrs[i] == nbytes for i in range(3))
    for i in (0, 1, 2, 5):
        np.testing.assert_array_equal(small[i], big[i][:3])
    again = call(10)
    for i in (0, 1, 2, 5):
        np.testing.assert_array_equal(again[i], big[i])
