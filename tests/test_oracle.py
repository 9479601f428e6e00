"""CPU tests of the oracle: pinned against NumPy/SciPy sub-oracles, against the golden fixtures
generated from the reference's own outer loop (tests/golden/make_goldens.py), and against
independent formulations (brute-force exact LTS, lstsq-literal FAST-LTS, closed-form plane wave).
"""
import json
import os

import numpy as np
import pytest
from scipy import signal

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False)


def _stdict_from_gold(g):
    if int(g['stdict_size']) < 0:
        return None
    d = {}
    off = 0
    for k, n in zip(g['stdict_keys'], g['stdict_lens']):
        d[str(k)] = g['stdict_vals'][off:off + n]
        off += n
    d['size'] = int(g['stdict_size'])
    return d


def test_vincenty_known_answer(oracle):
    # Geoscience Australia's GRS80 test line Flinders Peak -> Buninyong: 54972.271 m,
    # 306 52' 05.37", 127 10' 25.07" (WGS84 differs from GRS80 by < 0.1 mm here)
    d, a12, a21 = oracle.vincenty_inverse(-(37 + 57 / 60 + 3.72030 / 3600), 144 + 25 / 60 + 29.52440 / 3600,
                                          -(37 + 39 / 60 + 10.15610 / 3600), 143 + 55 / 60 + 35.38390 / 3600)
    assert abs(d - 54972.271) < 2e-3
    assert abs(a12 - (306 + 52 / 60 + 5.37 / 3600)) < 1e-6
    assert abs(a21 - (127 + 10 / 60 + 25.07 / 3600)) < 1e-6


def test_get_rij_matches_reference_golden(oracle):
    g = json.load(open(os.path.join(GOLD, 'planners.json')))['get_rij']
    rij = oracle.get_rij(g['lat'], g['lon'], 4)
    np.testing.assert_allclose(rij, np.array(g['rij']), rtol=0, atol=1e-12)
    assert abs(rij.mean(axis=1)).max() < 1e-12
    with pytest.raises(ValueError):
        oracle.get_rij([1.0], [1.0, 2.0], 2)


def test_filter_is_scipy_sosfilt_and_obspy_recipe(oracle):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 5000))
    st = oracle.make_stream(x, 20.0)
    stf, fs, sos = oracle.filter_data(st, 'cheby1', 0.5, 2.0, 2, 0.01)
    ref_sos = signal.iirfilter(2, [0.5, 2.0], rp=0.01, btype='band', analog=False, ftype='cheby1', fs=20.0, output='sos')
    np.testing.assert_array_equal(sos, ref_sos)
    assert sos.shape == (2, 6)
    tap = oracle.taper_window(5000)
    assert tap[0] == 0.0 and tap[50] == 1.0 and tap[-1] == 0.0 and np.count_nonzero(tap != 1.0) == 100
    np.testing.assert_array_equal(stf[1].data, signal.sosfilt(ref_sos, x[1]) * tap)
    stb, _, sosb = oracle.filter_data(st, 'butter', 0.5, 2.0, 2, 0.01)
    z, p, k = signal.iirfilter(2, [0.5 / 10, 2.0 / 10], btype='band', ftype='butter', output='zpk')
    s2 = signal.zpk2sos(z, p, k)
    y = signal.sosfilt(s2, signal.sosfilt(s2, x[2])[::-1])[::-1] * tap
    np.testing.assert_array_equal(stb[2].data, y)
    np.testing.assert_array_equal(sosb, signal.iirfilter(2, [0.5, 2.0], btype='band', ftype='butter', fs=20.0, output='sos'))
    np.testing.assert_array_equal(st[0].data, x[0])          # input untouched


def test_correlate_is_np_correlate_argmax_nanmedian(oracle):
    rng = np.random.default_rng(3)
    W, N = 64, 4
    data = rng.standard_normal((200, N))
    data[:, 3] = 0.0                                         # dead channel -> NaN column
    idx = oracle.pair_table(N)
    tau, mdccm, cmax = oracle.correlate_windows(data, W, np.array([0, 50]), idx, 10.0)
    for jj, t0 in enumerate((0, 50)):
        vals = []
        for k, (i, j) in enumerate(idx):
            a, b = data[t0:t0 + W, i], data[t0:t0 + W, j]
            with np.errstate(invalid='ignore'):
                c = np.correlate(a, b, "full") / np.sqrt(np.sum(a * a) * np.sum(b * b))
            assert tau[k, jj] == (W - (np.argmax(c) + 1)) / 10.0
            vals.append(c.max())
            if j == 3:
                assert np.isnan(cmax[k, jj]) and tau[k, jj] == (W - 1) / 10.0
        assert mdccm[jj] == np.nanmedian(vals)
    # sign convention: element j receives the signal d samples AFTER element i -> tau_ij = +d/fs
    s = rng.standard_normal(300)
    two = np.stack((s[20:220], s[15:215], s[10:210]), axis=1)   # ch1 lags ch0 by 5, ch2 by 10
    tau, _, _ = oracle.correlate_windows(two, 200, np.array([0]), oracle.pair_table(3), 1.0)
    np.testing.assert_array_equal(tau[:, 0], [5.0, 10.0, 5.0])


def test_ols_is_lstsq(oracle):
    rng = np.random.default_rng(5)
    rij = rng.standard_normal((2, 6))
    xij, idx = oracle.co_array(rij)
    tau = np.round(rng.standard_normal((15, 7)) * 20) / 20.0
    z, vel, baz, sig = oracle.ols_solve(xij, tau)
    zl = np.linalg.lstsq(xij, tau, rcond=None)[0]
    np.testing.assert_allclose(z, zl, rtol=1e-11, atol=1e-13)
    r = tau - xij @ zl
    np.testing.assert_allclose(sig, np.sqrt(np.sum(tau * r, axis=0) / 13), rtol=1e-9)
    np.testing.assert_allclose(vel, 1 / np.linalg.norm(zl, axis=0), rtol=1e-11)
    assert np.all((baz >= 0) & (baz < 360))


def test_noise_free_plane_wave_closed_form(oracle):
    """Integer-sample delays of a broadband signal: lags equal the analytic ones and OLS returns
    the slowness of the quantised delays exactly."""
    rng = np.random.default_rng(11)
    fs = 20.0
    rij = np.array([[0.0, 0.5, -0.4, 0.3, -0.2], [0.0, 0.3, 0.45, -0.5, -0.35]])
    baz, vel = np.radians(60.0), 0.33
    u = -np.array([np.sin(baz), np.cos(baz)])
    d = np.rint((u @ rij) / vel * fs).astype(int)
    s = rng.standard_normal(4000)
    data = np.stack([s[200 - di:3200 - di] for di in d])
    st = oracle.make_stream(data, fs)
    out, internals = oracle.ltsva(st, None, None, 30.0, 0.5, 1.0, rij=rij, return_internals=True)
    want = np.array([d[j] - d[i] for (i, j) in internals['idx_pair']]) / fs
    assert np.all(internals['tau'] == want[:, None])
    zq = np.linalg.lstsq(internals['xij'], want, rcond=None)[0]
    np.testing.assert_allclose(out[0], 1 / np.linalg.norm(zq), rtol=1e-10)
    np.testing.assert_allclose(out[1], np.degrees(np.arctan2(zq[0], zq[1])) % 360, rtol=1e-10)
    assert np.all(out[3] > 0.9) and np.all(out[3] <= 1.0)
    assert abs(out[1][0] - 60.0) < 3.0 and abs(out[0][0] - vel) < 0.03
    W = internals['W']
    np.testing.assert_array_equal(out[2], oracle.times_matplotlib(st[0])[internals['intervals'] + W // 2])


def test_lts_h_and_starts(oracle):
    assert [oracle.lts_h(P, a) for P, a in ((15, 0.75), (28, 0.5), (120, 0.5), (496, 0.5))] == [12, 15, 61, 249]
    rng = np.random.default_rng(2)
    xs = rng.standard_normal((15, 2))
    st = oracle.lts_starts(xs)
    assert st.shape == (105, 4) and np.all(st[:, 2:] == -1)
    xs = rng.standard_normal((120, 2))
    st = oracle.lts_starts(xs)
    assert st.shape == (500, 4)
    sub = oracle.uniran_subsets(120)
    assert np.all(sub[:, 0] != sub[:, 1]) and sub.min() >= 0 and sub.max() < 120
    seed = (0 * 5761 + 999) % 65536
    assert sub[0, 0] == int(seed / 65536.0 * 120)
    # a collinear 2-subset is extended to rank 2
    xs = np.array([[1.0, 0.0], [2.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
    st = oracle.lts_starts(xs)
    assert list(st[0][:3]) == [0, 1, 2]


def test_fast_lts_against_literal_and_brute_force(oracle):
    rng = np.random.default_rng(9)
    rij = rng.uniform(-1, 1, size=(2, 6))
    xij, idx = oracle.co_array(rij)
    ztrue = np.array([-2.0, 1.5])
    fs = 20.0
    for trial in range(6):
        tau = np.rint((xij @ ztrue + 0.02 * rng.standard_normal(15)) * fs) / fs
        bad = [k for k, (i, j) in enumerate(idx) if 5 in (i, j)]
        tau[bad] += rng.choice([-1, 1]) * 0.5
        z = oracle.fast_lts(tau[:, None], xij, 0.5)[:, 0]
        zl = oracle.fast_lts_literal(tau, xij, 0.5)
        np.testing.assert_allclose(z, zl, rtol=1e-9, atol=1e-12)
        h = oracle.lts_h(15, 0.5)
        r = tau - xij @ z
        obj = np.sum(np.sort(r * r)[:h])
        best, zb = oracle.brute_force_lts_objective(tau, xij, 0.5)
        # FAST-LTS works on MAD-standardised data; its optimum in original units is within a hair of exact LTS
        assert obj <= best * (1 + 1e-6) + 1e-12 or np.allclose(z, zb, rtol=1e-6)
        zf, wts, sig = oracle.lts_post_process(tau[:, None], xij, z[:, None], 0.5)
        assert set(np.where(wts[:, 0] == 0)[0]) == set(bad)
        good = [k for k in range(15) if k not in bad]
        np.testing.assert_allclose(zf[:, 0], np.linalg.lstsq(xij[good], tau[good], rcond=None)[0], rtol=1e-9)


def test_lts_degenerate_inputs(oracle):
    rng = np.random.default_rng(4)
    xij, idx = oracle.co_array(rng.uniform(-1, 1, size=(2, 6)))
    tau = np.zeros((15, 2))
    tau[:3, 1] = 0.05                                         # MAD(tau) == 0 in both columns
    z = oracle.fast_lts(tau, xij, 0.5)
    assert np.all(np.isnan(z))
    zf, wts, sig = oracle.lts_post_process(tau, xij, z, 0.5)
    assert np.all(np.isnan(zf)) and np.all(wts == 1) and np.all(np.isnan(sig))


@pytest.mark.parametrize('name', ['loop_ols_cheby1_adaptive', 'loop_ols_butter_linear',
                                  'loop_lts_butter_octave', 'loop_lts_2octave'])
def test_oracle_band_loop_equals_reference_loop(oracle, name, capsys):
    """The oracle's restatement of the band loop against the outputs of the REFERENCE's own
    narrow_band_least_squares()/..._parallel() (run with ltsva/obspy stubbed by the oracle)."""
    g = _load(name)
    st = oracle.make_stream(g['data'], float(g['fs']), starttime=17884.0729166667)
    nb = len(g['num_compute'])
    fr = g['freq_resp']
    w = np.zeros(len(fr))
    out = oracle.narrow_band_least_squares(list(g['winlens']), 0.5, float(g['alpha']), st, None, None, nb, w, w,
                                           list(g['freqlist']), str(g['band_type']), fr, str(g['ftype']), 2, 0.01,
                                           rij=g['rij'])
    capsys.readouterr()
    assert out[6] == list(g['num_compute'])
    assert out[0].shape == (nb, int(g['vector_len']))
    for i, key in ((0, 'vel'), (1, 'baz'), (2, 'mdccm'), (3, 't'), (5, 'sig')):
        np.testing.assert_allclose(out[i], g[key], rtol=1e-9, atol=1e-15, err_msg=key)
    np.testing.assert_allclose(out[7], g['w_array'], rtol=1e-12)
    np.testing.assert_allclose(out[8], g['h_array'], rtol=1e-9, atol=1e-13)   # golden made with scipy 1.7
    sd = _stdict_from_gold(g)
    if sd is None:
        assert out[4] is None
    else:
        assert set(out[4].keys()) == set(sd.keys())
        for k in sd:
            np.testing.assert_array_equal(out[4][k], sd[k])
        # every LTS fixture exercises the reference's key-prefix code on a NON-EMPTY dictionary, prefixes of several bands
        # (the overlapping-band fixture used to hold no entry at all)
        prefixes = {k.split('_')[0] for k in sd if k != 'size'}
        assert len(sd) > 10 and len(prefixes) >= 2, (name, len(sd), prefixes)


def test_oracle_filter_matches_the_references_own_filter_data(oracle):
    """tests/golden/filter_cheby1_ref.npz = output of the REFERENCE's filter_data (helpers.py:108-141, cheby1 branch:
    SciPy design + causal sosfilt per trace + 1 % taper) run in this container (make_goldens.py filter): the oracle's
    restatement reproduces it exactly."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'filter_cheby1_ref.npz'), allow_pickle=False)
    st = oracle.make_stream(g['data'], float(g['fs']))
    stf, fs, sos = oracle.filter_data(st, 'cheby1', float(g['fmin']), float(g['fmax']), int(g['order']), float(g['ripple']))
    np.testing.assert_array_equal(sos, g['sos'])
    np.testing.assert_array_equal(np.array([tr.data for tr in stf]), g['filtered'])
