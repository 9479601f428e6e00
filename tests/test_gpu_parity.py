"""GPU parity tests: the HIP path (through the ctypes C ABI) against the CPU oracle on the same
seeded inputs.  Tolerances: lags / weights / dropped elements exact; filtered samples 1e-11 of
the trace scale; MdCCM 1e-12 relative; vel / baz / sigma_tau 1e-9 relative (north_star asks 1e-6).
"""
import numpy as np
import pytest

from narrow_band_least_squares_amd import (engine, synthetic, filter_data, ltsva, narrow_band_least_squares,
                                           narrow_band_loop, get_rij)
from narrow_band_least_squares_amd import _hip

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _cfg(name, scale):
    return synthetic.build_config(name, scale=scale)


def _ostream(oracle, c):
    return oracle.make_stream(c['data'], c['fs'], starttime=c['st'][0].stats.starttime)


def test_library_loads_and_device_opens():
    h = engine.get_handle()
    assert h.lib.nbls_version() >= 100


def test_mfma_f64_layout():
    """A[i][k]: lane = i + 16k; B[k][j]: lane = j + 16k; D[i][j] at lane = j + 16*(i%4), reg = i//4."""
    rng = np.random.default_rng(1)
    A = rng.integers(-8, 8, size=(16, 4)).astype(float)
    B = rng.integers(-8, 8, size=(4, 16)).astype(float)
    a = np.zeros(64); b = np.zeros(64)
    for lane in range(64):
        a[lane] = A[lane & 15, lane >> 4]
        b[lane] = B[lane >> 4, lane & 15]
    out = engine.get_handle().probe_mfma_f64(a, b)
    D = A @ B
    got = np.zeros((16, 16))
    for lane in range(64):
        for reg in range(4):
            got[(lane >> 4) + 4 * reg, lane & 15] = out[lane, reg]
    np.testing.assert_array_equal(got, D)


def test_mfma_i8_layout():
    """i8 16x16x64: A row = lane & 15, B col = lane & 15, lane's 16 bytes are K-group lane >> 4;
    D[i][j] at lane = j + 16*(i//4), reg = i % 4 (the i32/f32 C/D map, unlike f64)."""
    rng = np.random.default_rng(2)
    A = rng.integers(-128, 128, size=(16, 64)).astype(np.int8)
    B = rng.integers(-128, 128, size=(64, 16)).astype(np.int8)
    a = np.zeros((64, 16), dtype=np.int8); b = np.zeros((64, 16), dtype=np.int8)
    for lane in range(64):
        a[lane] = A[lane & 15, 16 * (lane >> 4):16 * (lane >> 4) + 16]
        b[lane] = B[16 * (lane >> 4):16 * (lane >> 4) + 16, lane & 15]
    out = engine.get_handle().probe_mfma_i8(a, b)
    D = A.astype(np.int64) @ B.astype(np.int64)
    got = np.zeros((16, 16), dtype=np.int64)
    for lane in range(64):
        for reg in range(4):
            got[4 * (lane >> 4) + reg, lane & 15] = out[lane, reg]
    np.testing.assert_array_equal(got, D)


def test_int8_screening_correlator_matches_fp64():
    """xcorr_impl=3 (int8 MFMA screening + FP64 verification) against the f64-MFMA kernel: identical
    lags on every (band, window, pair), maxima to rounding; including a dead channel (all lags tie)."""
    c = _cfg('cfg3', 0.03)
    data, fs, t0 = engine.stream_to_array(c['st'])
    data = data.copy()
    data[3, 5000:9000] = 0.0                    # some windows see a dead element
    edges = [(0.1, 0.12), (0.2, 0.5), (1.0, 3.0), (4.0, 8.0)]
    wl = [30.0, 30.0, 20.0, 10.0]
    kw = dict(want_lag=True, want_cmax=True)
    r64 = engine.process(data, fs, t0, c['rij'], edges, wl, 0.5, 0.5, 'butter', 2, 0.01, xcorr_impl=2, **kw)
    r8 = engine.process(data, fs, t0, c['rij'], edges, wl, 0.5, 0.5, 'butter', 2, 0.01, xcorr_impl=3, **kw)
    np.testing.assert_array_equal(r8.lag, r64.lag)
    np.testing.assert_allclose(r8.cmax, r64.cmax, rtol=1e-12, atol=1e-15)
    for k in ('vel', 'baz', 'weights'):
        np.testing.assert_array_equal(getattr(r8, k), getattr(r64, k))
    c6 = _cfg('cfg2', 0.1)
    d6, fs6, t6 = engine.stream_to_array(c6['st'])
    a = engine.process(d6, fs6, t6, c6['rij'], [(0.5, 2.0)], [30.0], 0.5, 1.0, 'cheby1', 2, 0.01, xcorr_impl=2, **kw)
    b = engine.process(d6, fs6, t6, c6['rij'], [(0.5, 2.0)], [30.0], 0.5, 1.0, 'cheby1', 2, 0.01, xcorr_impl=3, **kw)
    np.testing.assert_array_equal(a.lag, b.lag)


@pytest.mark.parametrize('ftype,fmin,fmax', [('butter', 0.5, 2.0), ('cheby1', 0.1, 0.5), ('butter', 0.1, 0.16)])
def test_filter_data_parity(oracle, ftype, fmin, fmax):
    c = _cfg('cfg1b', 1.0)
    stf_o, fs_o, sos_o = oracle.filter_data(_ostream(oracle, c), ftype, fmin, fmax, 2, 0.01)
    stf, fs, sos = filter_data(c['st'], ftype, fmin, fmax, 2, 0.01)
    assert fs == fs_o
    np.testing.assert_array_equal(sos, sos_o)
    scale = max(np.abs(tr.data).max() for tr in stf_o)
    for a, b in zip(stf, stf_o):
        assert len(a.data) == len(b.data)
        assert np.max(np.abs(a.data - b.data)) <= 1e-11 * scale
    # the input stream is not modified
    np.testing.assert_array_equal(c['st'][0].data, c['data'][0])


@pytest.mark.parametrize('ftype,order', [('butter', 1), ('butter', 3), ('butter', 4), ('cheby1', 5), ('butter', 8)])
def test_filter_orders_cover_every_state_kernel(oracle, ftype, order):
    """1..8 second-order sections: state sizes 2, 8 and 16 take the matrix-core state kernel, 6 and 10 the
    VALU one; zero-phase (butter) fuses the backward states into the forward apply."""
    c = _cfg('cfg2', 0.3)
    stf, fs, sos = filter_data(c['st'], ftype, 0.8, 3.0, order, 0.01)
    stf_o, _, sos_o = oracle.filter_data(_ostream(oracle, c), ftype, 0.8, 3.0, order, 0.01)
    np.testing.assert_array_equal(sos, sos_o)
    scale = max(np.abs(tr.data).max() for tr in stf_o)
    for a, b in zip(stf, stf_o):
        assert np.max(np.abs(a.data - b.data)) <= 1e-11 * scale


def _compare_ltsva(oracle, c, stf_o, winlen, alpha):
    out_o, internals = oracle.ltsva(stf_o, None, None, winlen, 0.5, alpha, rij=c['rij'], return_internals=True)
    data = np.array([tr.data for tr in stf_o])
    res = engine.process(data, c['fs'], oracle.start_datenum(stf_o[0].stats.starttime), c['rij'],
                         [(None, None)], [winlen], 0.5, alpha, prefiltered=True, want_lag=True,
                         want_cmax=True, want_z=True, want_uncert=True)
    n = int(res.nwin[0])
    assert n == len(out_o[0])
    # confidence intervals (uncertainty_kernel, behind the solve): the same closed form evaluated by the oracle on the
    # SAME z and sigma_tau -> rounding only; end to end (the oracle's own z / sigma_tau, its dense-sampling evaluation) below
    cv_x, cb_x = oracle.confidence_intervals_closed_form(internals['xij'], res.z[0, :n].T, res.sigma_tau[0, :n])
    np.testing.assert_allclose(res.vel_uncert[0, :n], cv_x, rtol=1e-9, atol=1e-15, equal_nan=True)
    np.testing.assert_allclose(res.baz_uncert[0, :n], cb_x, rtol=1e-9, atol=1e-10, equal_nan=True)
    assert not np.any(res.vel_uncert[0, n:]) and not np.any(res.baz_uncert[0, n:])
    # (the oracle's ltsva samples the ellipse at 20 000 points: its own discretisation error reaches 1e-5 for the huge
    #  ellipses of incoherent windows whose boundary passes close to the origin)
    np.testing.assert_allclose(res.vel_uncert[0, :n], out_o[6], rtol=1e-4, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(res.baz_uncert[0, :n], out_o[7], rtol=1e-4, atol=1e-7, equal_nan=True)
    lag_o = np.rint(internals['tau'].T * c['fs']).astype(int)
    np.testing.assert_array_equal(res.lag[0, :n], lag_o)
    np.testing.assert_allclose(res.cmax[0, :n], internals['cmax'].T, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(res.mdccm[0, :n], out_o[3], rtol=1e-12)
    np.testing.assert_array_equal(res.t[0, :n], out_o[2])
    np.testing.assert_allclose(res.z[0, :n], internals['z'].T, rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(res.vel[0, :n], out_o[0], rtol=RTOL)
    np.testing.assert_allclose(res.baz[0, :n], out_o[1], rtol=RTOL)
    np.testing.assert_allclose(res.sigma_tau[0, :n], out_o[5], rtol=1e-7, atol=1e-12, equal_nan=True)
    if alpha < 1.0:
        np.testing.assert_array_equal(res.weights[0, :n], internals['weights'].T)
    return out_o


@pytest.mark.parametrize('alpha', [1.0, 0.75, 0.5])
def test_ltsva_parity(oracle, alpha):
    c = _cfg('cfg2', 0.25)
    stf_o, _, _ = oracle.filter_data(_ostream(oracle, c), 'butter', 0.5, 2.0, 2, 0.01)
    out_o = _compare_ltsva(oracle, c, stf_o, 30.0, alpha)
    # the public entry point, with the drop-in 8-tuple
    st = synthetic.make_stream(np.array([tr.data for tr in stf_o]), c['fs'], starttime=c['st'][0].stats.starttime)
    vel, baz, t, mdccm, stdict, sig, cv, cb = ltsva(st, None, None, 30.0, 0.5, alpha, False, rij=c['rij'])
    np.testing.assert_allclose(vel, out_o[0], rtol=RTOL)
    np.testing.assert_allclose(baz, out_o[1], rtol=RTOL)
    np.testing.assert_array_equal(t, out_o[2])
    assert set(stdict.keys()) == set(out_o[4].keys())
    for k in stdict:
        np.testing.assert_array_equal(stdict[k], out_o[4][k])
    assert len(cv) == len(vel) and len(cb) == len(vel)
    np.testing.assert_allclose(cv, out_o[6], rtol=1e-5, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(cb, out_o[7], rtol=1e-5, atol=1e-9, equal_nan=True)
    assert np.isfinite(cv).all()


@pytest.mark.parametrize('alpha', [0.5, 0.55, 0.9])
def test_ltsva_eight_elements_lts(oracle, alpha):
    c = _cfg('cfg3', 0.02)     # 8 elements, 28 pairs, 378 starts; h = 15, 16, 26
    stf_o, _, _ = oracle.filter_data(_ostream(oracle, c), 'butter', 1.0, 4.0, 2, 0.01)
    _compare_ltsva(oracle, c, stf_o, 30.0, alpha)


@pytest.mark.parametrize('nchans,alpha', [(4, 0.5), (5, 0.6), (6, 0.9), (7, 0.5)])
def test_ltsva_small_arrays_register_kernel(oracle, nchans, alpha):
    """4..7 elements (6, 10, 15, 21 pairs): every instantiation of the register-resident LTS kernel (its own
    sorting network each) against the oracle."""
    fs, npts = 20.0, 4000
    rij = synthetic.array_geometry(nchans, 1.0, seed=40 + nchans)
    data = synthetic.plane_wave(rij, npts, fs, 0.5, 4.0, timing_error_s=0.25, bad_element=nchans - 1, seed=21 + nchans)
    st = synthetic.make_stream(data, fs)
    c = dict(rij=rij - rij.mean(axis=1, keepdims=True), fs=fs, data=data, st=st)
    stf_o, _, _ = oracle.filter_data(_ostream(oracle, c), 'butter', 0.5, 4.0, 2, 0.01)
    _compare_ltsva(oracle, c, stf_o, 30.0, alpha)


@pytest.mark.parametrize('alpha,nine', [(0.5, False), (0.75, False), (0.5, True)])
def test_ltsva_regular_grid_ties_and_singular_starts(oracle, alpha, nine):
    """3x3 grid without its centre: many pairs share one baseline vector, so their residuals tie
    EXACTLY whenever their lags agree (the h-subset then hangs on the stable index order: the
    rank-counting path of the LTS kernel), and elemental starts made of two such pairs are singular
    (non-finite fit).  Everything must still equal the oracle: lags, z, weights, dropped elements."""
    fs, npts = 20.0, 6000
    rij = np.array([(x, y) for x in (0.0, 0.3, 0.6) for y in (0.0, 0.3, 0.6)]).T
    if not nine:
        rij = rij[:, [0, 1, 2, 3, 5, 6, 7, 8]]      # 8 elements: register kernel; 9: the large-array kernel
    data = synthetic.plane_wave(rij, npts, fs, 0.5, 4.0, timing_error_s=0.25, bad_element=rij.shape[1] - 1, seed=7)
    st = synthetic.make_stream(data, fs)
    c = dict(rij=rij - rij.mean(axis=1, keepdims=True), fs=fs, data=data, st=st)
    stf_o, _, _ = oracle.filter_data(_ostream(oracle, c), 'butter', 0.5, 4.0, 2, 0.01)
    out_o, internals = oracle.ltsva(stf_o, None, None, 30.0, 0.5, alpha, rij=c['rij'], return_internals=True)
    lag = np.rint(internals['tau'] * fs).astype(int)
    assert max(len(col) - len(set(col)) for col in lag.T) >= 8          # the ties are really there
    _compare_ltsva(oracle, c, stf_o, 30.0, alpha)


@pytest.mark.parametrize('keep,alpha', [([0, 1, 3, 4], 0.5), ([0, 1, 2, 3, 5], 0.5), ([0, 1, 2, 3, 4, 5], 0.75),
                                        ([0, 1, 2, 3, 5, 6, 8], 0.5)])
def test_ltsva_small_regular_grids_take_the_tie_path(oracle, keep, alpha):
    """4, 5, 6 and 7 elements on a regular grid: equal |r| across position h in every instantiation of the
    register-resident LTS kernel (subset = the values below the h-th smallest plus the FIRST tied ones in index
    order; csrc/solve.hip, select_reg)."""
    fs, npts = 20.0, 5000
    grid = np.array([(x, y) for x in (0.0, 0.3, 0.6) for y in (0.0, 0.3, 0.6)]).T
    rij = grid[:, keep]
    data = synthetic.plane_wave(rij, npts, fs, 0.5, 4.0, timing_error_s=0.25, bad_element=rij.shape[1] - 1, seed=11 + len(keep))
    st = synthetic.make_stream(data, fs)
    c = dict(rij=rij - rij.mean(axis=1, keepdims=True), fs=fs, data=data, st=st)
    stf_o, _, _ = oracle.filter_data(_ostream(oracle, c), 'butter', 0.5, 4.0, 2, 0.01)
    _compare_ltsva(oracle, c, stf_o, 30.0, alpha)


@pytest.mark.parametrize('nchans,alpha', [(9, 0.5), (12, 0.75), (16, 0.5), (20, 0.75), (32, 0.5)])
def test_ltsva_large_arrays_bucket_kernel(oracle, monkeypatch, nchans, alpha):
    """9..32 elements (36..496 pairs, 500 random starts): the large-array LTS kernel (solve_bucket.inc; VERDICT r02
    calls this test ..._cooperative_kernel after the kernel it replaced) against the oracle, and against the generic
    lane-per-start kernel (option "lts_impl" = 1) bit for bit."""
    fs, npts = 20.0, 3000
    rij = synthetic.array_geometry(nchans, 1.5)
    data = synthetic.plane_wave(rij, npts, fs, 0.5, 4.0, timing_error_s=0.25, bad_element=nchans - 1, seed=11)
    st = synthetic.make_stream(data, fs)
    c = dict(rij=rij - rij.mean(axis=1, keepdims=True), fs=fs, data=data, st=st)
    stf_o, _, _ = oracle.filter_data(_ostream(oracle, c), 'butter', 0.5, 4.0, 2, 0.01)
    _compare_ltsva(oracle, c, stf_o, 30.0, alpha)
    filt = np.array([tr.data for tr in stf_o])
    t0 = oracle.start_datenum(stf_o[0].stats.starttime)
    kw = dict(prefiltered=True, want_z=True)
    a = engine.process(filt, fs, t0, c['rij'], [(None, None)], [30.0], 0.5, alpha, **kw)
    h = engine.get_handle()
    h.set_option('lts_impl', 1)
    try:
        b = engine.process(filt, fs, t0, c['rij'], [(None, None)], [30.0], 0.5, alpha, **kw)
    finally:
        h.set_option('lts_impl', 0)
    np.testing.assert_array_equal(a.z, b.z)
    np.testing.assert_array_equal(a.weights, b.weights)
    np.testing.assert_array_equal(a.sigma_tau, b.sigma_tau)


def _compare_nbls(oracle, c, freq_resp, **kw):
    w = np.zeros(len(freq_resp)); h = np.zeros(len(freq_resp))
    args = (c['WINLEN_list'], c['overlap'], c['alpha'], None, None, None, c['NBANDS'], w, h, c['freqlist'],
            c['band_type'], freq_resp, c['ftype'], c['order'], c['ripple'])
    a = list(args); a[3] = c['st']
    got = narrow_band_least_squares(*a, rij=c['rij'])
    a[3] = _ostream(oracle, c)
    exp = oracle.narrow_band_least_squares(*a, rij=c['rij'])
    assert got[6] == exp[6]
    for i, name in ((0, 'vel'), (1, 'baz'), (2, 'mdccm')):
        np.testing.assert_allclose(got[i], exp[i], rtol=RTOL, atol=0, err_msg=name)
    np.testing.assert_array_equal(got[3], exp[3])
    np.testing.assert_allclose(got[5], exp[5], rtol=1e-7, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(got[7], exp[7], rtol=1e-13)
    np.testing.assert_allclose(got[8], exp[8], rtol=1e-12, atol=1e-300)
    if c['alpha'] == 1.0:
        assert got[4] is None and exp[4] is None
    else:
        assert set(got[4].keys()) == set(exp[4].keys())
        for k in got[4]:
            np.testing.assert_array_equal(got[4][k], exp[4][k])
    return got, exp


def test_narrow_band_cfg1_ols_butter(oracle):
    c = _cfg('cfg1', 1.0)
    _compare_nbls(oracle, c, np.logspace(-2, 1, 200))


def test_narrow_band_cfg1b_example_parameters(oracle):
    """example.py's literal parameters: 8 log bands 0.1-5 Hz, cheby1/2/0.01, adaptive 60->30 s."""
    c = _cfg('cfg1b', 1.0)
    got, exp = _compare_nbls(oracle, c, np.logspace(-2, 1, 1000))
    assert got[6] == [39, 42, 46, 50, 56, 62, 69, 79]
    assert got[0].shape == (8, 80)


def test_narrow_band_cfg2_lts(oracle):
    c = _cfg('cfg2', 0.2)
    c['alpha'] = 0.5
    got, exp = _compare_nbls(oracle, c, np.logspace(-2, 1, 100))
    assert any(k != 'size' for k in got[4])


def test_narrow_band_loop_matches_batched_row(oracle):
    c = _cfg('cfg1', 0.5)
    fr = np.logspace(-2, 1, 50)
    w = np.zeros(50)
    full = narrow_band_least_squares(c['WINLEN_list'], 0.5, 1.0, c['st'], None, None, c['NBANDS'], w, w,
                                     c['freqlist'], 'linear', fr, 'butter', 2, 0.01, rij=c['rij'])
    vl = full[0].shape[1]
    row = narrow_band_loop(3, c['freqlist'], 'linear', fr, c['st'], 'butter', 2, 0.01, None, None,
                           c['WINLEN_list'], 0.5, 1.0, vl, rij=c['rij'])
    np.testing.assert_array_equal(row[0], full[0][3])
    np.testing.assert_array_equal(row[1], full[1][3])
    np.testing.assert_array_equal(row[2], full[2][3])
    np.testing.assert_array_equal(row[6], full[5][3])
    assert int(row[7]) == full[6][3]
    assert row[4] is None and row[5] is None


def test_kernel_variants_agree(monkeypatch):
    """The f64-MFMA correlator against the plain VALU one (lags identical, maxima to rounding) and the
    register-resident LTS kernel against the generic one (bit identical)."""
    c = _cfg('cfg3', 0.03)
    data, fs, t0 = engine.stream_to_array(c['st'])
    edges = [(0.2, 0.5), (1.0, 3.0), (4.0, 8.0)]
    kw = dict(want_lag=True, want_cmax=True, want_z=True)
    r_mfma = engine.process(data, fs, t0, c['rij'], edges, [30.0, 20.0, 10.0], 0.5, 0.5, 'butter', 2, 0.01, xcorr_impl=2, **kw)
    r_valu = engine.process(data, fs, t0, c['rij'], edges, [30.0, 20.0, 10.0], 0.5, 0.5, 'butter', 2, 0.01, xcorr_impl=1, **kw)
    np.testing.assert_array_equal(r_mfma.lag, r_valu.lag)
    np.testing.assert_allclose(r_mfma.cmax, r_valu.cmax, rtol=1e-12, atol=1e-15)
    for k in ('vel', 'baz', 'sigma_tau', 'weights', 'z'):
        np.testing.assert_array_equal(getattr(r_mfma, k), getattr(r_valu, k))
    h = engine.get_handle()
    h.set_option('lts_impl', 1)
    try:
        r_gen = engine.process(data, fs, t0, c['rij'], edges, [30.0, 20.0, 10.0], 0.5, 0.5, 'butter', 2, 0.01, xcorr_impl=2, **kw)
        # the other identical-result switches of the library, all at once
        for key in ('screen_tb4', 'screen_nsl1', 'screen_static', 'screen_pretest', 'filter_nofuse', 'filter_nomfma', 'lts_generic_h'):
            h.set_option(key, 1)
        h.set_option('lts_impl', 0)
        r_alt = engine.process(data, fs, t0, c['rij'], edges, [30.0, 20.0, 10.0], 0.5, 0.5, 'butter', 2, 0.01, xcorr_impl=3, **kw)
    finally:
        for key in ('lts_impl', 'screen_tb4', 'screen_nsl1', 'screen_static', 'screen_pretest', 'filter_nofuse', 'filter_nomfma', 'lts_generic_h'):
            h.set_option(key, 0)
    np.testing.assert_array_equal(r_alt.lag, r_mfma.lag)
    for k in ('vel', 'baz', 'weights'):
        np.testing.assert_array_equal(getattr(r_alt, k), getattr(r_mfma, k))
    if not h.lib.nbls_developer_build():          # timing switches that falsify results are not in the shipped library
        with pytest.raises(ValueError):
            h.set_option('ablate', 1)
    with pytest.raises(ValueError):
        h.set_option('no_such_option', 1)
    for k in ('vel', 'baz', 'sigma_tau', 'weights', 'z', 'mdccm'):
        np.testing.assert_array_equal(getattr(r_mfma, k), getattr(r_gen, k))


@pytest.mark.parametrize('swap', [False, True])
def test_two_raw_maxima_that_share_one_quotient_go_to_the_earlier_lag(oracle, swap):
    """the reference divides the correlation by the norm and THEN takes the first maximum
    (np.argmax(cij / norm)); two raw values 1 ulp apart can share one quotient, and the earlier lag then wins although
    its raw value is the smaller one.  Crafted so that every operation is exact on both sides: channel a holds M - 1 and M
    (M just below 2^53, chosen so that (M - 1) / norm == M / norm in double), channel b a single 1.0 — the correlation IS
    a.  All three correlators must pick the oracle's lag (they used to pick the raw maximum: 21 samples later)."""
    M = 9007199254740988.0
    W, fs = 64, 20.0
    rng = np.random.default_rng(5)
    a = np.zeros(W + 1)
    b = np.zeros(W + 1)
    # (np.correlate(x_i, x_j) runs forward through x_i and backward through x_j: the smaller value must come first in
    #  index order either way)
    a[20], a[41], b[5] = (M, M - 1.0, 1.0) if swap else (M - 1.0, M, 1.0)
    nrm = np.sqrt(np.sum(a[:W] * a[:W]) * np.sum(b[:W] * b[:W]))
    assert (M - 1.0) / nrm == M / nrm and M - 1.0 < M          # the premise: one quotient, two raw values
    data = rng.standard_normal((4, W + 1))
    data[0], data[1] = (b, a) if swap else (a, b)
    rij = synthetic.array_geometry(4, 1.0, seed=3)
    rij = rij - rij.mean(axis=1, keepdims=True)
    st = oracle.make_stream(data, fs)
    out, internals = oracle.ltsva(st, None, None, W / fs, 0.5, 1.0, rij=rij, return_internals=True)
    lag_o = np.rint(internals['tau'].T * fs).astype(int)
    assert lag_o.shape == (1, 6)
    full = np.correlate(data[0, :W], data[1, :W], 'full')
    assert (W - 1) - lag_o[0, 0] == np.argmax(full / nrm) != np.argmax(full)      # pair (0, 1): the earlier of the two
    for impl in (1, 2, 3):
        res = engine.process(data, fs, 0.0, rij, [(None, None)], [W / fs], 0.5, 1.0, prefiltered=True, want_lag=True,
                             want_cmax=True, xcorr_impl=impl)
        np.testing.assert_array_equal(res.lag[0, :1], lag_o, err_msg='xcorr_impl %d' % impl)
        np.testing.assert_allclose(res.cmax[0, :1], internals['cmax'].T, rtol=1e-12)


def test_resident_trace_skips_the_upload_and_changes_nothing(monkeypatch):
    """``with engine.resident_trace(st)``: the samples go up once; calls on the same buffers inside the block skip their
    upload (counted here on the handle), give the results of ordinary calls bit for bit — several parameter sets, the
    band-group form of rounds 2-3 included — and a different trace in between ends the residency instead of being
    processed on the old samples."""
    from narrow_band_least_squares_amd import resident_trace
    c = _cfg('cfg2', 0.25)
    fr = np.logspace(-2, 1, 24)
    w = np.zeros(24)

    def args(nb, alpha, st):
        return (c['WINLEN_list'][:nb], 0.5, alpha, st, None, None, nb, w, w, c['freqlist'][:nb + 1], 'log', fr, 'butter', 2, 0.01)

    def same(a, b):
        for i in (0, 1, 2, 3, 5, 7, 8):
            np.testing.assert_array_equal(a[i], b[i])
        assert a[6] == b[6]
        if a[4] is None:
            assert b[4] is None
        else:
            assert list(a[4].keys()) == list(b[4].keys())
            for k in a[4]:
                np.testing.assert_array_equal(a[4][k], b[4][k])

    plain = {(nb, al): narrow_band_least_squares(*args(nb, al, c['st']), rij=c['rij']) for nb, al in ((6, 0.5), (9, 1.0))}
    h = engine.get_handle()
    uploads = []
    for name in ('upload_rows', 'set_trace_rows', 'set_trace'):
        real = getattr(h, name)
        monkeypatch.setattr(h, name, lambda *a, _r=real, _n=name, **k: (uploads.append(_n), _r(*a, **k))[1])
    with resident_trace(c['st']):
        assert len(uploads) == 1
        for key in plain:
            same(narrow_band_least_squares(*args(*key, c['st']), rij=c['rij']), plain[key])
        monkeypatch.setenv('NBLS_PIPELINE_GROUPS', '3')
        same(narrow_band_least_squares(*args(6, 0.5, c['st']), rij=c['rij']), plain[(6, 0.5)])
        monkeypatch.delenv('NBLS_PIPELINE_GROUPS')
        assert len(uploads) == 1, uploads
        # another trace on the same GPU: processed from ITS samples, and the first one has to go up again afterwards
        other = synthetic.make_stream(np.array([tr.data for tr in c['st']])[:, ::-1].copy(), c['fs'], starttime=c['st'][0].stats.starttime)
        rev = narrow_band_least_squares(*args(6, 0.5, other), rij=c['rij'])
        assert len(uploads) == 2 and not np.array_equal(rev[1], plain[(6, 0.5)][1])
        same(narrow_band_least_squares(*args(6, 0.5, c['st']), rij=c['rij']), plain[(6, 0.5)])
        assert len(uploads) == 3
    same(narrow_band_least_squares(*args(6, 0.5, c['st']), rij=c['rij']), plain[(6, 0.5)])
    assert len(uploads) == 4


@pytest.mark.parametrize('alpha', [0.5, 1.0])
def test_streamed_pass_equals_the_unstreamed_one(alpha, monkeypatch):
    """The default whole call since round 4 (nbls_stream_results): every unit batch a complete correlate -> solve -> pack
    chain whose rows reach a pinned host mirror while the next batch runs.  With a screening batch small enough for a
    dozen batches that cut through bands: rows, lags, weights and the dictionary bit-identical to the pass fetched in one
    piece, to the band groups of rounds 2-3, and to the solve on the second stream (option "overlap")."""
    c = _cfg('cfg2', 0.25)
    fr = np.logspace(-2, 1, 24)
    w = np.zeros(24)
    nb = 9
    args = (c['WINLEN_list'][:nb], 0.5, alpha, c['st'], None, None, nb, w, w, c['freqlist'][:nb + 1], 'log', fr, 'butter', 2, 0.01)
    h = engine.get_handle()
    monkeypatch.delenv('NBLS_PIPELINE_GROUPS', raising=False)
    monkeypatch.setenv('NBLS_STREAM_RESULTS', '0')
    whole = narrow_band_least_squares(*args, rij=c['rij'])
    monkeypatch.setenv('NBLS_PIPELINE_GROUPS', '3')
    grouped = narrow_band_least_squares(*args, rij=c['rij'])
    monkeypatch.delenv('NBLS_PIPELINE_GROUPS')
    monkeypatch.setenv('NBLS_STREAM_RESULTS', '1')      # (a call of this size is fetched in one piece by itself: engine.stream_pays)
    outs = {}
    try:
        h.set_option('screen_batch_mb', 1)
        h.set_option('solve_min_units', 1)           # (default: small screening batches are solved several at a time)
        outs['streamed'] = narrow_band_least_squares(*args, rij=c['rij'])
        nbatch = h.result_batches()
        assert nbatch >= 3, nbatch
        h.set_option('overlap', -1)                  # (0 = auto: the solves of a several-batch streamed pass go to the second stream)
        outs['streamed, solves behind their batch on one stream'] = narrow_band_least_squares(*args, rij=c['rij'])
        h.set_option('overlap', 1)
        outs['overlap'] = narrow_band_least_squares(*args, rij=c['rij'])
        # the engine-level entry with the side arrays (fetched after the last batch)
        data, fs, t0 = engine.stream_to_array(c['st'])
        edges = [(c['freqlist'][b], c['freqlist'][b + 1]) for b in range(nb)]
        kw = dict(want_lag=True, want_cmax=True, want_z=True)
        seen = []
        r_s = engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'][:nb], 0.5, alpha, 'butter', 2, 0.01,
                             units_done=lambda res, u0, u1: seen.append((u0, u1)), **kw)
        h.set_option('overlap', -1)
        h.set_option('screen_batch_mb', 192)
        r_w = engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'][:nb], 0.5, alpha, 'butter', 2, 0.01, groups=1, **kw)
    finally:
        h.set_option('overlap', 0)
        h.set_option('screen_batch_mb', 192)
        h.set_option('solve_min_units', 0)
    assert len(seen) >= 3 and seen[0][0] == 0 and seen[-1][1] == int(r_s.nwin.sum())
    assert all(a[1] == b[0] for a, b in zip(seen[:-1], seen[1:]))              # consecutive, in order
    for k in ('vel', 'baz', 'mdccm', 'sigma_tau', 'mask', 'lag', 'cmax', 'z', 't'):
        np.testing.assert_array_equal(getattr(r_s, k), getattr(r_w, k), err_msg=k)
    for name, got in list(outs.items()) + [('grouped', grouped)]:
        assert got[6] == whole[6]
        for i in (0, 1, 2, 3, 5, 7, 8):
            np.testing.assert_array_equal(got[i], whole[i], err_msg='%s[%d]' % (name, i))
        if alpha == 1.0:
            assert got[4] is None
        else:
            assert list(got[4].keys()) == list(whole[4].keys()), name
            for k in whole[4]:
                np.testing.assert_array_equal(got[4][k], whole[4][k])


def test_streamed_batches_through_the_c_abi():
    """nbls_stream_results / nbls_result_batches / nbls_wait_result_batch as a C caller uses them: the cells of a batch in
    the pinned mirror equal the same cells of nbls_fetch_packed, batch by batch; a pass on a general correlator (no unit
    batches) is ONE batch; without nbls_stream_results there is nothing to wait for."""
    c = _cfg('cfg2', 0.2)
    data, fs, t0 = engine.stream_to_array(c['st'])
    nb = 6
    edges = [(c['freqlist'][b], c['freqlist'][b + 1]) for b in range(nb)]
    prep = engine.prepare(data.shape[0], data.shape[1], fs, c['rij'], edges, c['WINLEN_list'][:nb], 0.5, 0.5, 'butter', 2, 0.01)
    h = engine.get_handle()
    try:
        h.set_option('screen_batch_mb', 1)
        h.set_option('solve_min_units', 1)
        engine.launch(h, data, prep, stream=True)
        n = h.result_batches()
        assert n >= 2
        got = [h.wait_result_batch(k) for k in range(n)]
        grids_m = got[-1][4].copy()
        mask_m = got[-1][5].copy()
        full = h.fetch_packed()
        fg = np.stack([full[k].reshape(-1) for k in ('vel', 'baz', 'mdccm', 'sigma_tau')])
        fm = full['mask'].reshape(-1, full['mask'].shape[2])
        U = int(prep.nwin.sum())
        assert got[0][0] == 0 and got[-1][1] == U
        covered = np.zeros(fg.shape[1], dtype=bool)
        for (u0, u1, c0, c1, _, _) in got:
            assert u1 > u0 and c1 - c0 >= u1 - u0
            covered[c0:c1] = True
            np.testing.assert_array_equal(grids_m[:, c0:c1], fg[:, c0:c1])
            np.testing.assert_array_equal(mask_m[c0:c1], fm[c0:c1])
        assert np.all(fg[:, ~covered] == 0.0)                    # what no batch covers is padding
        # general correlator: one batch with everything
        engine.launch(h, data, prep, xcorr_impl=1, stream=True)
        assert h.result_batches() == 1
        u0, u1, c0, c1, g1, m1 = h.wait_result_batch(0)
        assert (u0, u1) == (0, U)
        g1 = g1.copy()
        full1 = h.fetch_packed()
        np.testing.assert_array_equal(g1[:, c0:c1], np.stack([full1[k].reshape(-1) for k in ('vel', 'baz', 'mdccm', 'sigma_tau')])[:, c0:c1])
        np.testing.assert_allclose(g1[:, c0:c1], fg[:, c0:c1], rtol=1e-12)        # (another correlator: maxima to rounding)
        engine.launch(h, data, prep)                             # not streamed
        assert h.result_batches() == 0
        with pytest.raises(_hip.NblsError):
            h.wait_result_batch(0)
        h.sync()
        # the side arrays are not cleared by a pass; a pass WITHOUT the correlation / solve stages must not hand
        # back the previous pass's lags and weights
        assert np.any(h.fetch(want_lag=True, grids=False)['lag'] != 0)
        h.execute(stages=1)
        side = h.fetch(want_lag=True, want_cmax=True, want_weights=True, want_z=True)
        for k in ('lag', 'cmax', 'weights', 'z', 'vel'):
            assert not np.any(side[k]), k
    finally:
        h.set_option('screen_batch_mb', 192)
        h.set_option('solve_min_units', 0)


@pytest.mark.parametrize('nchans,winlen', [(3, 20.0), (4, 12.5), (5, 30.0), (7, 9.0), (9, 25.0), (12, 15.0), (16, 40.0),
                                           (17, 10.0), (20, 12.0), (32, 15.0), (6, 12.525), (8, 12.175), (8, 51.3), (20, 60.0), (12, 70.0)])
def test_correlators_agree_for_any_array_size(nchans, winlen):
    """Tile geometry depends on the element count (lag blocks per tile, partner skew, idle columns):
    the int8-screening and f64-MFMA correlators must pick the lags of the plain VALU kernel for every
    array size they accept, including window lengths that are not multiples of the tile steps."""
    rij = synthetic.array_geometry(nchans, 1.0, seed=100 + nchans)
    data = synthetic.plane_wave(rij, 6000, 40.0, 0.2, 8.0, seed=7 + nchans)
    kw = dict(want_lag=True, want_cmax=True)
    edges = [(0.3, 2.0), (2.0, 6.0)]
    ref = engine.process(data, 40.0, 0.0, rij, edges, [winlen, winlen], 0.5, 1.0, 'cheby1', 2, 0.01, xcorr_impl=1, **kw)
    for impl in (2, 3):
        try:
            got = engine.process(data, 40.0, 0.0, rij, edges, [winlen, winlen], 0.5, 1.0, 'cheby1', 2, 0.01,
                                 xcorr_impl=impl, **kw)
        except _hip.NblsError:
            # the f64-MFMA correlator takes up to 16 elements and window sets that fit LDS
            assert impl == 2 and (nchans > 16 or nchans * winlen * 40.0 * 8 > 150e3)
            continue
        np.testing.assert_array_equal(got.lag, ref.lag, err_msg='impl %d' % impl)
        np.testing.assert_allclose(got.cmax, ref.cmax, rtol=1e-12, atol=1e-15)
        np.testing.assert_array_equal(got.baz, ref.baz)


@pytest.mark.parametrize('nchans', [3, 4, 5, 6, 7, 9, 13, 17, 19])
def test_screening_on_incoherent_noise_for_any_array_size(nchans):
    """On white noise the arg-max of a pair lies anywhere among the 2W-1 lags, so every lag block of every tile column
    (8 blocks per column with three elements, 5 with four, ... 1 from ten on) has to be right — a plane wave, whose maxima
    sit at small lags, does not notice a wrong column mapping of the far blocks."""
    rng = np.random.default_rng(400 + nchans)
    rij = synthetic.array_geometry(nchans, 1.0, seed=300 + nchans)
    data = rng.standard_normal((nchans, 5200))
    kw = dict(want_lag=True, want_cmax=True)
    edges = [(0.5, 4.0), (4.0, 12.0)]
    ref = engine.process(data, 40.0, 0.0, rij, edges, [17.3, 9.0], 0.5, 1.0, 'butter', 2, 0.01, xcorr_impl=1, **kw)
    got = engine.process(data, 40.0, 0.0, rij, edges, [17.3, 9.0], 0.5, 1.0, 'butter', 2, 0.01, xcorr_impl=3, **kw)
    assert np.abs(ref.lag).max() > 200                      # the far lag blocks are exercised
    np.testing.assert_array_equal(got.lag, ref.lag)
    np.testing.assert_allclose(got.cmax, ref.cmax, rtol=1e-12, atol=1e-15)


def test_band_passes_when_hbm_budget_is_small(monkeypatch):
    """More bands than the filtered-trace budget allows are run in consecutive passes: same rows."""
    c = _cfg('cfg1', 0.3)
    data, fs, t0 = engine.stream_to_array(c['st'])
    edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(c['NBANDS'])]
    full = engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], 0.5, 1.0, 'butter', 2, 0.01)
    monkeypatch.setenv('NBLS_MAX_FILTERED_GB', str(3.5 * 8 * data.size / 2.0 ** 30))     # three bands per pass
    assert engine.max_bands_per_pass(*data.shape) == 3
    split = engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], 0.5, 1.0, 'butter', 2, 0.01)
    for k in ('vel', 'baz', 'mdccm', 'sigma_tau', 't', 'nwin'):
        np.testing.assert_array_equal(getattr(split, k), getattr(full, k))
    assert len(split.sos) == len(edges)


def test_lts_whole_call_in_several_hbm_rounds(monkeypatch):
    """ALPHA < 1 with more bands than the filtered-trace budget of one pass: the rounds land inside the
    launch loop, before the key text exists — tuple, dictionary and key ORDER must equal the one-pass call's."""
    c = _cfg('cfg2', 0.2)
    fr = np.logspace(-2, 1, 32)
    w = np.zeros(32)
    nb = 7
    args = (c['WINLEN_list'][:nb], 0.5, 0.5, c['st'], None, None, nb, w, w, c['freqlist'][:nb + 1], c['band_type'], fr, 'butter', 2, 0.01)
    one = narrow_band_least_squares(*args, rij=c['rij'])
    data = c['data']
    monkeypatch.setenv('NBLS_MAX_FILTERED_GB', repr(3.5 * 8 * data.size / 2.0 ** 30))          # three bands per pass
    assert engine.max_bands_per_pass(*data.shape) == 3
    rounds = narrow_band_least_squares(*args, rij=c['rij'])
    for i in (0, 1, 2, 3, 5, 7, 8):
        np.testing.assert_array_equal(rounds[i], one[i])
    assert rounds[6] == one[6] and list(rounds[4].keys()) == list(one[4].keys())
    assert any(k != 'size' for k in one[4])
    for k in one[4]:
        np.testing.assert_array_equal(rounds[4][k], one[4][k])


def test_pipelined_band_groups_equal_one_pass(monkeypatch):
    """The whole call cut into 1, 2 and 4 band groups (concurrent passes on several handles of the same GPU,
    dictionary built group by group): identical tuples, identical key order."""
    c = _cfg('cfg2', 0.3)
    fr = np.logspace(-2, 1, 32)
    w = np.zeros(32)
    args = (c['WINLEN_list'], 0.5, 0.5, c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr,
            'butter', 2, 0.01)
    outs = []
    for g in ('1', '2', '4'):
        monkeypatch.setenv('NBLS_PIPELINE_GROUPS', g)
        outs.append(narrow_band_least_squares(*args, rij=c['rij']))
    for o in outs[1:]:
        for i in (0, 1, 2, 3, 5, 7, 8):
            np.testing.assert_array_equal(o[i], outs[0][i])
        assert o[6] == outs[0][6] and list(o[4].keys()) == list(outs[0][4].keys())
        for k in outs[0][4]:
            np.testing.assert_array_equal(o[4][k], outs[0][4][k])


def test_stream_priority_and_upload_overlap_do_not_change_results(monkeypatch):
    """The groups' handles run on prioritised streams (slot 0 highest) and the trace goes up on a helper thread while
    the first filters are designed: scheduling only.  Every priority value is accepted (clamped to the device's
    range), and the call gives the same tuple with both mechanisms switched off."""
    c = _cfg('cfg2', 0.3)
    fr = np.logspace(-2, 1, 32)
    w = np.zeros(32)
    args = (c['WINLEN_list'], 0.5, 0.5, c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr,
            'butter', 2, 0.01)
    monkeypatch.setenv('NBLS_PIPELINE_GROUPS', '3')
    ref = narrow_band_least_squares(*args, rij=c['rij'])
    h = engine.get_handle(None, 1)
    try:
        for prio in (-5, 7, 1, 0):
            h.set_option('stream_priority', prio)
        monkeypatch.setattr(engine, 'UPLOAD_OVERLAP', False)
        monkeypatch.setattr(engine, 'GROUP_ORDER', False)        # the groups' passes not ordered on the GPU either
        out = narrow_band_least_squares(*args, rij=c['rij'])
    finally:
        h.set_option('stream_priority', 0)
    for i in (0, 1, 2, 3, 5, 7, 8):
        np.testing.assert_array_equal(out[i], ref[i])
    assert out[6] == ref[6] and list(out[4].keys()) == list(ref[4].keys())
    for k in ref[4]:
        np.testing.assert_array_equal(out[4][k], ref[4][k])
    # nbls_execute_after: a handle cannot wait for itself, nor for a handle that has never queued a pass
    from narrow_band_least_squares_amd._hip import Handle, NblsError
    fresh = Handle(engine.default_device())
    try:
        with pytest.raises((NblsError, ValueError, RuntimeError)):
            h.execute(after=fresh)
        with pytest.raises((NblsError, ValueError, RuntimeError)):
            h.execute(after=h)
    finally:
        fresh.close()


def test_two_step_trace_upload_states():
    """nbls_set_trace_shape + nbls_upload_rows: a pass cannot be queued before the samples are there, rows that do not
    match the declared shape are refused, and the two-step form gives the rows of the one-step form."""
    from narrow_band_least_squares_amd._hip import Handle, NblsError
    c = _cfg('cfg2', 0.1)
    data, fs, t0 = engine.stream_to_array(c['st'])
    rows = [np.ascontiguousarray(r) for r in data]
    edges = [(0.5, 1.0), (1.0, 2.0)]
    ref = engine.process(data, fs, t0, c['rij'], edges, [30.0, 30.0], 0.5, 0.5, 'butter', 2, 0.01)
    prep = engine.prepare(len(rows), len(rows[0]), fs, c['rij'], edges, [30.0, 30.0], 0.5, 0.5, 'butter', 2, 0.01)
    h = Handle(engine.default_device())
    try:
        h.set_trace_shape(len(rows), len(rows[0]), fs)
        with pytest.raises((NblsError, ValueError, RuntimeError)):          # planned, but no samples yet
            engine.launch(h, rows, prep, trace_ready=True)
        with pytest.raises(ValueError):
            h.upload_rows(rows[:-1])                                        # a row is missing
        with pytest.raises(ValueError):
            h.upload_rows([r[:-1] for r in rows])                           # wrong length
        h.upload_rows(rows)
        engine.launch(h, rows, prep, trace_ready=True)
        out = h.fetch_packed()
        for k in ('vel', 'baz', 'mdccm'):
            np.testing.assert_array_equal(out[k], getattr(ref, k))
        np.testing.assert_array_equal(out['mask'], ref.mask)
    finally:
        h.close()


def test_pass_queued_while_the_rows_are_still_going_up():
    """nbls_expect_upload: the pass is planned and queued while nbls_upload_rows runs (or has not even begun) on another
    thread; the filter stage takes the channels as their events are recorded.  Results equal the one-step form bit for
    bit — also with the filter forced into launches of one and of three channels on a trace that is there already
    (option filter_row_step: every (band, channel) series is independent), zero-phase and causal filters; an aborted
    or failed upload ends the waiting pass with an error at once, and a declared trace nobody announced is refused."""
    import threading
    import time
    from narrow_band_least_squares_amd._hip import Handle, NblsError
    c = _cfg('cfg2', 0.25)
    data, fs, t0 = engine.stream_to_array(c['st'])
    rows = [np.ascontiguousarray(r) for r in data]
    edges = [(0.3, 0.6), (0.6, 1.2), (1.2, 2.4)]
    wl = [40.0, 30.0, 20.0]
    for ftype, alpha in (('butter', 0.5), ('cheby1', 1.0)):
        prep = engine.prepare(len(rows), len(rows[0]), fs, c['rij'], edges, wl, 0.5, alpha, ftype, 2, 0.01)
        h = Handle(engine.default_device())
        try:
            engine.launch(h, rows, prep)
            ref = h.fetch_packed()
            ref_filt = [h.fetch_filtered(b) for b in range(len(edges))]
            for step in (1, 3):
                h.set_option('filter_row_step', step)
                engine.launch(h, rows, prep)
                out = h.fetch_packed()
                for k in ref:
                    np.testing.assert_array_equal(out[k], ref[k])
                for b in range(len(edges)):
                    np.testing.assert_array_equal(h.fetch_filtered(b), ref_filt[b])
            h.set_option('filter_row_step', 0)
            for delay in (0.0, 0.05):
                h.set_trace_shape(len(rows), len(rows[0]), fs)
                h.expect_upload()

                def up():
                    time.sleep(delay)
                    h.upload_rows(rows)
                th = threading.Thread(target=up)
                th.start()
                engine.launch(h, rows, prep, trace_ready=True)      # set_geometry, plan, execute: the rows may not be there yet
                th.join()
                out = h.fetch_packed()
                for k in ref:
                    np.testing.assert_array_equal(out[k], ref[k])
            # the announced rows never come: the announcing side says so, the pass fails at once
            h.set_trace_shape(len(rows), len(rows[0]), fs)
            h.expect_upload()
            th = threading.Thread(target=lambda: (time.sleep(0.05), h.lib.nbls_abort_upload(h._h)))
            th.start()
            t_ = time.perf_counter()
            with pytest.raises((NblsError, RuntimeError, ValueError)):
                engine.launch(h, rows, prep, trace_ready=True)
            th.join()
            assert time.perf_counter() - t_ < 5.0
            # rows of the wrong length: refused by upload_rows, which also releases a pass that would wait for them
            h.set_trace_shape(len(rows), len(rows[0]), fs)
            h.expect_upload()
            with pytest.raises(ValueError):
                h.upload_rows([r[:-1] for r in rows])
            with pytest.raises((NblsError, RuntimeError, ValueError)):
                engine.launch(h, rows, prep, trace_ready=True)
            # declared, not announced: no waiting
            h.set_trace_shape(len(rows), len(rows[0]), fs)
            t_ = time.perf_counter()
            with pytest.raises((NblsError, RuntimeError, ValueError)):
                engine.launch(h, rows, prep, trace_ready=True)
            assert time.perf_counter() - t_ < 5.0
            h.upload_rows(rows)
            with pytest.raises((NblsError, RuntimeError, ValueError)):
                h.expect_upload()                                    # (nothing is declared any more: the samples are there)
            engine.launch(h, rows, prep, trace_ready=True)
            out = h.fetch_packed()
            for k in ref:
                np.testing.assert_array_equal(out[k], ref[k])
        finally:
            h.close()


def test_upload_error_surfaces_on_the_calling_thread():
    """The helper thread's exception (rows of unequal length) is re-raised by the call, not lost."""
    c = _cfg('cfg2', 0.1)
    data, fs, t0 = engine.stream_to_array(c['st'])
    rows = [np.ascontiguousarray(r) for r in data]
    rows[2] = rows[2][:-5]
    edges = [(0.5, 1.0), (1.0, 2.0)]
    with pytest.raises(ValueError):
        engine.process(rows, fs, t0, c['rij'], edges, [30.0, 30.0], 0.5, 1.0, 'butter', 2, 0.01)
    # the handle is still usable
    ok = engine.process(data, fs, t0, c['rij'], edges, [30.0, 30.0], 0.5, 1.0, 'butter', 2, 0.01)
    assert np.isfinite(ok.vel[:, :int(ok.nwin[0])]).all()


def test_window_slices_add_up_to_the_full_run():
    """Window sharding (fewer bands than GPUs): the slices of the windows processed separately are
    disjoint, keep their global row index and add up to the unsliced run bit for bit."""
    c = _cfg('cfg1b', 1.0)          # adaptive windows: every band has its own count
    data, fs, t0 = engine.stream_to_array(c['st'])
    edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(c['NBANDS'])]
    kw = dict(vector_len=80)
    full = engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], 0.5, 0.5, 'cheby1', 2, 0.01, **kw)
    parts = [engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], 0.5, 0.5, 'cheby1', 2, 0.01,
                            window_slice=(k, 3), **kw) for k in range(3)]
    for name in ('vel', 'baz', 'mdccm', 'sigma_tau'):
        np.testing.assert_array_equal(sum(getattr(p, name) for p in parts), getattr(full, name))
    np.testing.assert_array_equal(sum(p.weights.astype(int) for p in parts), full.weights.astype(int))
    assert all(np.array_equal(p.nwin, full.nwin) and np.array_equal(p.t, full.t) for p in parts)
    assert np.count_nonzero(parts[1].vel[0]) < np.count_nonzero(full.vel[0])
    after = engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], 0.5, 0.5, 'cheby1', 2, 0.01, **kw)
    np.testing.assert_array_equal(after.vel, full.vel)          # the slice setting does not leak


@pytest.mark.parametrize('mode', ['all', 'rank'])
def test_band_sharded_entry_point_under_rccl(mode):
    """narrow_band_least_squares_parallel() through the library's own RCCL gather (nbls_comm_*; no
    PyTorch, no launcher): both ways of forming the communicator with this box's one GPU, bands- and
    windows-sharded paths forced; must equal the serial call bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'tests', '_dist_gpu_worker.py'), mode]
    r = subprocess.run(cmd, env=dict(os.environ), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'DIST_GPU_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def _loopback_env():
    """Environment of a process whose ranks are several handles on this box's one GPU, with the loopback stand-in
    (tests/c_caller/loopback_rccl.cpp, built here with hipcc) where RCCL would be."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, 'tests', 'c_caller', 'loopback_rccl.cpp')
    lib = os.path.join(root, 'tests', 'c_caller', 'libloopback_rccl.so')
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.run(['/opt/rocm/bin/hipcc', '-O2', '-shared', '-fPIC', '--offload-arch=gfx950', src, '-o', lib], check=True, timeout=300)
    env = dict(os.environ)
    env.update(NBLS_TEST_TRANSPORT=lib)        # read by tests/_dist_gpu_worker.py; bench.py takes --transport-lib
    return root, env


@pytest.mark.parametrize('nranks', [2, 3, 8])
def test_several_ranks_of_one_process_on_one_gpu_over_a_loopback_transport(nranks):
    """What a one-GPU box can run of the multi-rank path: narrow_band_least_squares_parallel() with 2, 3 and 8 ranks of ONE
    process (a handle, a launch thread and a result block in HBM per rank; LPT band shares and window slices; a share in
    several HBM rounds), nbls_comm_gather with its status words to root 0, the root's assembly — the blocks moved by a
    loopback stand-in instead of RCCL.  Equal to the serial call bit for bit."""
    import os
    import subprocess
    import sys
    root, env = _loopback_env()
    env['NBLS_DEVICES'] = ','.join(['0'] * nranks)
    cmd = [sys.executable, os.path.join(root, 'tests', '_dist_gpu_worker.py'), 'loop%d' % nranks]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'DIST_GPU_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_drives_two_ranks_without_a_launcher_over_the_loopback_transport():
    """`python bench.py --gpus 2` exactly as the driver starts it (no launcher): the one-process form, here with both
    ranks on this box's one GPU and the loopback stand-in — a functional rehearsal of the bench's multi-GPU code path
    (its numbers mean nothing: two ranks share one GPU)."""
    import json
    import os
    import subprocess
    import sys
    root, env = _loopback_env()
    env['NBLS_DEVICES'] = '0,0'
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--config', 'cfg2', '--steps', '2', '--warmup', '1',
           '--no-cpu-baseline', '--no-noise', '--transport-lib', env['NBLS_TEST_TRANSPORT']]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['value'] and line['value'] > 0
    assert 'one process drives all of them' in line['config']['parallelism']


def _launch_two_ranks(script_args, env, port, timeout=600):
    import subprocess
    import sys
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port)] + script_args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_one_process_per_rank_under_the_launcher_on_one_gpu_over_the_loopback_transport():
    """The launcher form the driver uses for its scaling runs, rehearsed with two ranks on this box's one GPU: RANK /
    WORLD_SIZE from torch.distributed.run, the communicator id over the TCP side channel, ncclCommInitRank, every rank
    computes its band share (or window slice) and the all-gather returns the complete result on every rank — equal to
    the serial call bit for bit on both.  (Stand-in transport: shared host memory.)"""
    import os
    root, env = _loopback_env()
    env['NBLS_DEVICE'] = '0'
    r = _launch_two_ranks([os.path.join(root, 'tests', '_dist_gpu_worker.py'), 'proc'], env, 29741)
    assert r.returncode == 0 and r.stdout.count('DIST_GPU_OK') == 2, r.stdout[-2000:] + r.stderr[-3000:]


def test_a_rank_that_fails_before_planning_does_not_hang_its_peer_on_hardware():
    """Launcher form, two ranks on this box's one GPU: rank 1 raises before it can plan.  It still takes part in the
    gather with an empty block whose status word is set (csrc/comm.hip), rank 0 raises "rank(s) [1] failed", and the
    next call on the same communicator is healthy and equal to the serial call."""
    import os
    root, env = _loopback_env()
    env['NBLS_DEVICE'] = '0'
    r = _launch_two_ranks([os.path.join(root, 'tests', '_dist_gpu_worker.py'), 'procfail'], env, 29747, timeout=300)
    assert r.returncode == 0 and r.stdout.count('DIST_GPU_OK') == 2, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_under_the_launcher_with_two_ranks_over_the_loopback_transport():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` — the command line of the driver's
    scaling run — with both ranks on this box's one GPU: barrier / max-over-ranks (gloo), the band-sharded whole call,
    the `independent_calls` leg, ONE JSON line from rank 0.  Functional rehearsal; the numbers mean nothing."""
    import json
    import os
    root, env = _loopback_env()
    env['NBLS_DEVICE'] = '0'
    r = _launch_two_ranks([os.path.join(root, 'bench.py'), '--gpus', '2', '--config', 'cfg2', '--steps', '2', '--warmup', '1',
                           '--no-noise', '--transport-lib', env['NBLS_TEST_TRANSPORT']], env, 29743)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['value'] and line['value'] > 0
    assert 'one process per GPU' in line['config']['parallelism']
    ind = line['independent_calls']
    assert ind['scaling'] == 'weak' and ind['n_gpus'] == 2 and ind['value'] > 0


@pytest.mark.parametrize('alpha', [0.75, 1.0])
def test_zero_edit_drop_in_route_with_lat_lon(oracle, tmp_path, alpha):
    """north_star: "drops into example.py".  tests/_dropin_script.py is written like the reference's script — its import
    lines verbatim after install_as_reference_modules(), lat/lon LISTS from tr.stats (Vincenty -> co-array on the
    way to the GPU; every other GPU test passes rij=), the positional PLOT_ARRAY_COORDINATES of the broadband ltsva
    call, example.py's literal parameters (cheby1, adaptive windows) — and must give the oracle's 9-tuple and
    broadband 8-tuple for the same lat/lon."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / 'dropin.npz'
    r = subprocess.run([sys.executable, os.path.join(root, 'tests', '_dropin_script.py'), str(out)],
                       env=dict(os.environ, DROPIN_ALPHA=repr(alpha)), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'DROPIN_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    g = np.load(out, allow_pickle=False)
    lat, lon = list(g['lat']), list(g['lon'])
    st = oracle.make_stream(g['data'], float(g['Fs']), starttime=17884.0729166667, lat=lat, lon=lon)
    np.testing.assert_allclose(g['rij'], oracle.get_rij(lat, lon, 8), rtol=0, atol=1e-12)
    # broadband: filter_data -> ltsva with lat/lon
    stf, fs, sos = oracle.filter_data(st, 'cheby1', 0.1, 5, 2, 0.01)
    exp = oracle.ltsva(stf, lat, lon, 50, 0.5, alpha, False)
    for i, k in ((0, 'vel_broad'), (1, 'baz_broad'), (3, 'mdccm_broad')):
        np.testing.assert_allclose(g[k], exp[i], rtol=RTOL, atol=0, err_msg=k)
    np.testing.assert_array_equal(g['t_broad'], exp[2])
    np.testing.assert_allclose(g['sig_tau_broad'], exp[5], rtol=1e-7, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(g['vel_uncert_broad'], exp[6], rtol=1e-5, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(g['baz_uncert_broad'], exp[7], rtol=1e-5, atol=1e-7, equal_nan=True)
    # narrow band
    args = (list(g['winlens']), 0.5, alpha, st, lat, lon, 8, g['w_broad'], g['h_broad'], list(g['freqlist']), 'log', g['freq_resp'],
            'cheby1', 2, 0.01)
    nb = oracle.narrow_band_least_squares(*args)
    assert list(g['num_compute']) == nb[6] == [39, 42, 46, 50, 56, 62, 69, 79]
    for i, k in ((0, 'vel'), (1, 'baz'), (2, 'mdccm')):
        np.testing.assert_allclose(g[k], nb[i], rtol=RTOL, atol=0, err_msg=k)
    np.testing.assert_array_equal(g['t'], nb[3])
    np.testing.assert_allclose(g['sig_tau'], nb[5], rtol=1e-7, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(g['h'], nb[8], rtol=1e-12, atol=1e-300)
    if alpha < 1.0:
        for tag, d in (('nb', nb[4]), ('bb', exp[4])):
            keys = [str(k) for k in g[tag + '_keys']]
            assert keys == [k for k in d if k != 'size'] and int(g[tag + '_size']) == d['size']
            off = 0
            for k, n in zip(keys, g[tag + '_lens']):
                np.testing.assert_array_equal(g[tag + '_vals'][off:off + n], d[k])
                off += n
        assert len(g['nb_keys']) > 0


@pytest.mark.parametrize('name', ['loop_ols_cheby1_adaptive', 'loop_ols_butter_linear', 'loop_lts_butter_octave',
                                  'loop_lts_2octave', 'loop_lts_101bands'])
def test_product_against_reference_loop_goldens(name):
    """The GPU path against the fixtures produced by the REFERENCE's own narrow_band_least_squares() /
    ..._parallel() loops (tests/golden/make_goldens.py; ltsva/obspy stubbed by the oracle there): shapes,
    num_compute_list, zero padding, stdict keys (incl. the overlapping '2_octave_over' bands, and the three-
    character '100_' / '101_' prefixes the reference's zfill(2) gives beyond 99 bands) and values."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.npz'), allow_pickle=False)
    st = synthetic.make_stream(g['data'], float(g['fs']), starttime=17884.0729166667)
    nb = len(g['num_compute'])
    fr = g['freq_resp']
    w = np.zeros(len(fr))
    out = narrow_band_least_squares(list(g['winlens']), 0.5, float(g['alpha']), st, None, None, nb, w, w,
                                    list(g['freqlist']), str(g['band_type']), fr, str(g['ftype']), 2, 0.01,
                                    rij=g['rij'])
    assert out[6] == list(g['num_compute']) and out[0].shape == (nb, int(g['vector_len']))
    for i, key in ((0, 'vel'), (1, 'baz'), (2, 'mdccm'), (5, 'sig')):
        np.testing.assert_allclose(out[i], g[key], rtol=1e-9, atol=1e-15, err_msg=key)
    np.testing.assert_array_equal(out[3], g['t'])
    np.testing.assert_allclose(out[8], g['h_array'], rtol=1e-9, atol=1e-13)
    if int(g['stdict_size']) < 0:
        assert out[4] is None
    else:
        keys = [str(k) for k in g['stdict_keys']]
        assert sorted(k for k in out[4] if k != 'size') == sorted(keys) and out[4]['size'] == int(g['stdict_size'])
        off = 0
        for k, n in zip(keys, g['stdict_lens']):
            np.testing.assert_array_equal(out[4][k], g['stdict_vals'][off:off + n])
            off += n


def test_zero_channel_nan_semantics(oracle):
    """An all-zero element: its pairs give 0/0 -> NaN maxima, argmax 0 (lag W-1), nanmedian skips them."""
    c = _cfg('cfg1', 0.25)
    data = c['data'].copy()
    data[2] = 0.0
    st_o = oracle.make_stream(data, c['fs'], starttime=c['st'][0].stats.starttime)
    _compare_ltsva(oracle, c, st_o, 30.0, 1.0)


@pytest.mark.parametrize('nchans,alpha', [(8, 0.5), (6, 0.75), (12, 0.5), (24, 0.75)])
def test_dead_channel_under_lts(oracle, nchans, alpha):
    """An all-zero element with LTS switched on: its N - 1 pairs carry NaN maxima and the lag of a 0/0 arg-max; the
    robust fit has to drop them (register kernel: 6 and 8 elements; large-array kernel: 12 and 24) exactly as the
    oracle does — lags, z, weights, dropped elements, MdCCM by nanmedian."""
    fs, npts = 20.0, 3600
    rij = synthetic.array_geometry(nchans, 1.2, seed=70 + nchans)
    data = synthetic.plane_wave(rij, npts, fs, 0.5, 4.0, seed=5 + nchans)
    data[nchans // 2] = 0.0
    c = dict(rij=rij - rij.mean(axis=1, keepdims=True), fs=fs, data=data, st=synthetic.make_stream(data, fs))
    stf_o, _, _ = oracle.filter_data(_ostream(oracle, c), 'butter', 0.5, 4.0, 2, 0.01)
    _compare_ltsva(oracle, c, stf_o, 30.0, alpha)


def test_errors_match_reference_types():
    c = _cfg('cfg1', 0.1)
    with pytest.raises(ValueError):
        get_rij([1.0, 2.0], [1.0], 2)
    st2 = synthetic.make_stream(c['data'][:2], c['fs'])
    with pytest.raises(RuntimeError):
        ltsva(st2, None, None, 30.0, 0.5, 1.0, rij=c['rij'][:, :2])
    st3 = synthetic.make_stream(c['data'][:3], c['fs'])
    with pytest.raises(RuntimeError):
        ltsva(st3, None, None, 30.0, 0.5, 0.75, rij=c['rij'][:, :3])
    line = np.vstack((np.arange(4.0), np.zeros(4)))
    st4 = synthetic.make_stream(c['data'][:4], c['fs'])
    with pytest.raises(RuntimeError):
        ltsva(st4, None, None, 30.0, 0.5, 1.0, rij=line)


def test_short_trace_gives_no_windows():
    c = _cfg('cfg1', 0.02)    # 480 samples < 600-sample window
    vel, baz, t, mdccm, stdict, sig, _, _ = ltsva(c['st'], None, None, 30.0, 0.5, 1.0, rij=c['rij'])
    assert len(vel) == 0 and len(t) == 0 and stdict == {}


def test_full_size_cfg3_properties():
    """BASELINE configs[2] at full size (48 bands x 1438 windows = 69 024 units): size-independent
    properties.  (i) broadband (one band 0.3-3 Hz over the same 6 h trace): the plane wave is recovered
    and the corrupted element is what LTS drops; (ii) the 48-band run is bit-reproducible; (iii) a band
    subset reproduces the same rows (what band sharding relies on); (iv) every unit is filled."""
    c = _cfg('cfg3', 1.0)
    data, fs, t0 = engine.stream_to_array(c['st'])
    rb = engine.process(data, fs, t0, c['rij'], [(0.3, 3.0)], [30.0], 0.5, 0.5, 'butter', 2, 0.01)
    n = int(rb.nwin[0])
    assert n == 1438
    assert abs(np.nanmedian(rb.baz[0, :n]) - 225.0) < 1.0
    assert abs(np.nanmedian(rb.vel[0, :n]) - 0.34) < 0.01
    dropped = rb.weights[0, :n] == 0
    bad_pairs = (rb.pair_idx[:, 0] == 7) | (rb.pair_idx[:, 1] == 7)
    assert dropped[:, bad_pairs].mean() > 0.9 and dropped[:, ~bad_pairs].mean() < 0.1
    edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(c['NBANDS'])]
    r1 = engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], 0.5, 0.5, 'butter', 2, 0.01)
    assert int(r1.nwin.sum()) == 69024 and np.all(r1.nwin == 1438)
    assert np.all(r1.mdccm > 0) and np.all(r1.mdccm <= 1.0 + 1e-12)
    assert np.isfinite(r1.vel).mean() > 0.999
    r2 = engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], 0.5, 0.5, 'butter', 2, 0.01)
    for k in ('vel', 'baz', 'mdccm', 'sigma_tau', 'weights'):
        np.testing.assert_array_equal(getattr(r1, k), getattr(r2, k))
    sub = [5, 24, 40]
    r3 = engine.process(data, fs, t0, c['rij'], [edges[i] for i in sub], [c['WINLEN_list'][i] for i in sub],
                        0.5, 0.5, 'butter', 2, 0.01, vector_len=r1.vel.shape[1])
    for j, i in enumerate(sub):
        for k in ('vel', 'baz', 'mdccm', 'weights'):
            np.testing.assert_array_equal(getattr(r3, k)[j], getattr(r1, k)[i])


def _full_size_properties(c, edges, winlens, nchans_bad, subset, broadband, check_tail_band=None):
    """Size-independent properties of a production-size pass (no oracle at this size): (i) bit-reproducible;
    (ii) a band subset reproduces the same rows (what band sharding relies on); (iii) every unit is filled;
    (iv) the mistimed element is what LTS drops and the plane wave is recovered (a broadband band over the same trace); (v) optionally: the LAST window of band ``check_tail_band`` equals a
    three-window run over the same filtered samples fed back as a pre-filtered trace — with 12 bands x 16 elements x
    8.64 M samples the filtered buffer holds 1.66e9 doubles, so this window lies beyond element 2^31 of it."""
    data, fs, t0 = engine.stream_to_array(c['st'])
    kw = dict(groups=1)
    r1 = engine.process(data, fs, t0, c['rij'], edges, winlens, 0.5, 0.5, 'butter', 2, 0.01, **kw)
    filt_tail = r1.handle.fetch_filtered(check_tail_band) if check_tail_band is not None else None
    n = r1.nwin.astype(int)
    tail = None
    if check_tail_band is not None:
        b = check_tail_band
        filt = filt_tail                                             # (N, npts) filtered + tapered band, as the device holds it
        filt_tail = None
        W, inc = int(r1.W[b]), int(r1.inc[b])
        s0 = (n[b] - 3) * inc
        tail = np.ascontiguousarray(filt[:, s0:s0 + 2 * inc + W + 1])
        del filt
    assert np.all(n > 0)
    for b in range(len(edges)):
        assert np.all(r1.mdccm[b, :n[b]] > 0) and np.all(r1.mdccm[b, :n[b]] <= 1.0 + 1e-12)
        assert np.isfinite(r1.vel[b, :n[b]]).mean() > 0.999
        assert np.all(r1.weights[b, :n[b]].sum(axis=1) >= r1.weights.shape[2] // 4)      # the reweighting keeps a majority-sized set
    # the mistimed element is what LTS drops, and the plane wave is recovered: one broadband band over the same trace
    # (the narrow bands have bandwidth x window length < 5: their correlation peaks are too broad to single it out)
    rb = engine.process(data, fs, t0, c['rij'], [broadband], [winlens[0]], 0.5, 0.5, 'butter', 2, 0.01, **kw)
    nbb = int(rb.nwin[0])
    assert nbb == n.max()
    assert abs(np.nanmedian(rb.baz[0, :nbb]) - 225.0) < 1.0 and abs(np.nanmedian(rb.vel[0, :nbb]) - 0.34) < 0.01
    dropped = rb.weights[0, :nbb] == 0
    bad_pairs = (rb.pair_idx[:, 0] == nchans_bad) | (rb.pair_idx[:, 1] == nchans_bad)
    assert dropped[:, bad_pairs].mean() > 0.9 and dropped[:, ~bad_pairs].mean() < 0.1
    r2 = engine.process(data, fs, t0, c['rij'], edges, winlens, 0.5, 0.5, 'butter', 2, 0.01, **kw)
    for k in ('vel', 'baz', 'mdccm', 'sigma_tau', 'weights'):
        np.testing.assert_array_equal(getattr(r1, k), getattr(r2, k))
    r3 = engine.process(data, fs, t0, c['rij'], [edges[i] for i in subset], [winlens[i] for i in subset], 0.5, 0.5, 'butter', 2, 0.01,
                        vector_len=r1.vel.shape[1], **kw)
    for j, i in enumerate(subset):
        for k in ('vel', 'baz', 'mdccm', 'weights'):
            np.testing.assert_array_equal(getattr(r3, k)[j], getattr(r1, k)[i])
    if tail is not None:
        b = check_tail_band
        rt = engine.process(tail, fs, t0, c['rij'], [(None, None)], [winlens[b]], 0.5, 0.5, prefiltered=True)
        assert int(rt.nwin[0]) == 3
        last = n[b] - 1
        for k in ('vel', 'baz', 'mdccm', 'weights', 'sigma_tau'):
            np.testing.assert_array_equal(getattr(rt, k)[0, 2], getattr(r1, k)[b, last], err_msg=k)
    return r1


def test_full_size_cfg4_share_properties(monkeypatch):
    """BASELINE configs[3] at full size, ONE GPU's share: band share 0 of 8 (12 of the 96 bands, the LPT partition of
    narrow_band_least_squares_parallel) of the 16-element, 24 h @ 100 Hz trace — 69 096 units, 120 pairs, 500 LCG
    starts, W = 3000."""
    monkeypatch.delenv('NBLS_MAX_FILTERED_GB', raising=False)     # ONE in-core pass holds all bands (the tail check reads one back)
    from narrow_band_least_squares_amd import dist
    c = _cfg('cfg4', 1.0)
    costs = dist.band_costs(c['npts'], c['fs'], list(c['WINLEN_list']), c['overlap'], 120)
    bands = dist.shard_bands(costs, 8)[0]
    assert len(bands) == 12
    edges = [(c['freqlist'][b], c['freqlist'][b + 1]) for b in bands]
    winlens = [c['WINLEN_list'][b] for b in bands]
    r = _full_size_properties(c, edges, winlens, 15, [1, 6, 11], (0.5, 4.0), check_tail_band=11)
    assert int(r.nwin.sum()) == 69096


def test_full_size_cfg5_properties(monkeypatch):
    """BASELINE configs[4] at full size: 32 elements (496 pairs, 500 LCG starts), all 128 bands, 1 h @ 20 Hz — 30 464
    units in one call (band prefixes beyond '99_' included in the dictionary path elsewhere)."""
    monkeypatch.delenv('NBLS_MAX_FILTERED_GB', raising=False)
    c = _cfg('cfg5', 1.0)
    edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(c['NBANDS'])]
    r = _full_size_properties(c, edges, list(c['WINLEN_list']), 31, [3, 64, 127], (0.3, 3.0), check_tail_band=127)
    assert int(r.nwin.sum()) == 30464 and len(edges) == 128


def test_full_size_cfg3_band_against_oracle(oracle):
    """One whole band of BASELINE configs[2] (the lowest, 0.100-0.110 Hz: broadest correlation peaks, most
    screening candidates; 1438 windows, 6 h @ 40 Hz) taken from the 48-band production run, against the
    oracle: lags and LTS weights exact, vel / baz / MdCCM to 1e-9."""
    c = _cfg('cfg3', 1.0)
    data, fs, t0 = engine.stream_to_array(c['st'])
    edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(c['NBANDS'])]
    res = engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], 0.5, 0.5, 'butter', 2, 0.01, want_lag=True)
    stf_o, _, _ = oracle.filter_data(_ostream(oracle, c), 'butter', edges[0][0], edges[0][1], 2, 0.01)
    out_o, internals = oracle.ltsva(stf_o, None, None, 30.0, 0.5, 0.5, rij=c['rij'], return_internals=True)
    n = int(res.nwin[0])
    assert n == len(out_o[0]) == 1438
    np.testing.assert_array_equal(res.lag[0, :n], np.rint(internals['tau'].T * fs).astype(int))
    np.testing.assert_array_equal(res.weights[0, :n], internals['weights'].T)
    np.testing.assert_allclose(res.vel[0, :n], out_o[0], rtol=RTOL)
    np.testing.assert_allclose(res.baz[0, :n], out_o[1], rtol=RTOL)
    np.testing.assert_allclose(res.mdccm[0, :n], out_o[3], rtol=1e-9)


@pytest.mark.parametrize('seed', range(16))
def test_random_configurations_against_oracle(oracle, seed):
    """Seeded random draws over what the drop-in call accepts: 4..9 elements, sampling rate, band type and
    count, window length (constant or adaptive), overlap, OLS or LTS with a random alpha, filter type and
    order, SNR, with or without a mistimed element — the whole 9-tuple against the oracle."""
    from narrow_band_least_squares_amd import helpers
    rng = np.random.default_rng(1000 + seed)
    nchans = int(rng.integers(4, 10))
    fs = float(rng.choice([20.0, 40.0, 50.0]))
    dur = float(rng.uniform(300.0, 700.0))
    npts = int(dur * fs) + int(rng.integers(0, 7))
    band_type = str(rng.choice(['linear', 'log', 'octave', '2_octave_over', 'onethird_octave']))
    fmin, fmax = float(rng.uniform(0.2, 0.6)), float(rng.uniform(3.0, 0.4 * fs))
    nb = int(rng.integers(2, 7))
    freqlist, nbands, _ = helpers.get_freqlist(fmin, fmax, band_type, nb)
    if rng.random() < 0.5:
        winlens = helpers.get_winlenlist('constant', nbands, 50, float(rng.choice([20.0, 30.0, 45.0])), 0)
    else:
        winlens = helpers.get_winlenlist('adaptive', nbands, 50, 60, 25)
    alpha = 1.0 if rng.random() < 0.35 else float(rng.choice([0.5, 0.6, 0.75, 0.9]))
    ftype = 'butter' if rng.random() < 0.6 else 'cheby1'
    bad = nchans - 1 if (alpha < 1.0 and rng.random() < 0.7) else None
    rij = synthetic.array_geometry(nchans, float(rng.uniform(0.5, 2.0)), seed=int(rng.integers(1 << 30)))
    data = synthetic.plane_wave(rij, npts, fs, fmin, min(fmax, 0.45 * fs), baz_deg=float(rng.uniform(0, 360)),
                                vel_kms=float(rng.uniform(0.3, 3.0)), snr_db=float(rng.uniform(-3, 12)),
                                timing_error_s=0.3 if bad is not None else 0.0, bad_element=bad,
                                seed=int(rng.integers(1 << 30)))
    c = dict(WINLEN_list=winlens, overlap=float(rng.choice([0.0, 0.25, 0.5, 0.75])), alpha=alpha,
             st=synthetic.make_stream(data, fs), NBANDS=nbands, freqlist=freqlist, band_type=band_type, ftype=ftype,
             order=int(rng.integers(1, 4)), ripple=0.01, rij=rij - rij.mean(axis=1, keepdims=True), data=data, fs=fs)
    _compare_nbls(oracle, c, np.logspace(-2, 1, 40))


@pytest.mark.parametrize('nchans,fs,winlen', [(6, 20.0, 30.0), (8, 20.0, 60.0), (8, 40.0, 30.0), (32, 20.0, 30.0)])
def test_baseline_shapes_use_the_screening_correlator(nchans, fs, winlen):
    """Array size x window length of every BASELINE configuration (cfg-4 has its own test below): the
    automatic choice is the int8 screening correlator (a fallback to the VALU kernel is 50x slower)."""
    rij = synthetic.array_geometry(nchans, 1.0, seed=nchans)
    data = synthetic.plane_wave(rij, int(2.6 * winlen * fs), fs, 0.5, 0.4 * fs, seed=3)
    h = engine.get_handle()
    h.set_profiling(True)
    try:
        engine.process(data, fs, 0.0, rij, [(0.5, 4.0)], [winlen], 0.5, 1.0, 'butter', 2, 0.01)
        assert h.timings()['xcorr_impl'] == 3
    finally:
        h.set_profiling(False)


def test_cfg4_window_shape_keeps_the_screening_correlator():
    """BASELINE configs[3] shape (16 elements, 30 s windows at 100 Hz = 3000 samples): the automatic
    choice must still be the int8 screening correlator (its LDS budget is tight there), with the lags of
    the plain VALU kernel."""
    rij = synthetic.array_geometry(16, 2.0, seed=5)
    data = synthetic.plane_wave(rij, 9000, 100.0, 0.5, 10.0, seed=6)
    kw = dict(want_lag=True, want_cmax=True)
    h = engine.get_handle()
    h.set_profiling(True)
    try:
        got = engine.process(data, 100.0, 0.0, rij, [(0.5, 5.0)], [30.0], 0.5, 1.0, 'butter', 2, 0.01, **kw)
        assert h.timings()['xcorr_impl'] == 3
    finally:
        h.set_profiling(False)
    ref = engine.process(data, 100.0, 0.0, rij, [(0.5, 5.0)], [30.0], 0.5, 1.0, 'butter', 2, 0.01, xcorr_impl=1, **kw)
    np.testing.assert_array_equal(got.lag, ref.lag)
    np.testing.assert_allclose(got.cmax, ref.cmax, rtol=1e-12, atol=1e-15)


def test_filter_responses_on_this_host_are_scipys_bit_for_bit():
    """The same check as tests/test_host.py's on the GPU box's CPU (another micro-architecture than the build
    container's): the w / h rows of a whole call are scipy.signal.sosfreqz's, bit for bit."""
    from scipy import signal
    c = _cfg('cfg2', 0.1)
    fr = np.logspace(-2, 1, 1000)
    w = np.zeros(len(fr))
    out = narrow_band_least_squares(c['WINLEN_list'], 0.5, 1.0, c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'],
                                    fr, 'cheby1', 2, 0.01, rij=c['rij'])
    fs = c['fs']
    for b in range(c['NBANDS']):
        sos = signal.iirfilter(2, [c['freqlist'][b], c['freqlist'][b + 1]], rp=0.01, btype='band', ftype='cheby1', output='sos', fs=fs)
        ws, hs = signal.sosfreqz(sos, fr, fs=fs)
        assert np.array_equal(np.asarray(out[7][b]).real.view(np.uint64), ws.view(np.uint64))
        assert np.array_equal(np.asarray(out[8][b]).view(np.uint64), hs.view(np.uint64))
