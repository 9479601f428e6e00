"""CPU tests of the host side: planners against the reference's goldens, the duck-typed stream,
text I/O, the C ABI's exported symbols, loud failure without a GPU, the LTS plan against the oracle's
constants, and the N>1 (band-sharded) path under gloo with world_size 2."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import narrow_band_least_squares_amd as pkg
from narrow_band_least_squares_amd import _hip, dist, planner, synthetic
from narrow_band_least_squares_amd.stream import Stream, Trace, Stats, start_datenum

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def test_planners_match_reference_goldens():
    g = json.load(open(os.path.join(GOLD, 'planners.json')))
    for kind in ('linear', 'log', 'octave', '2_octave_over', 'onethird_octave', 'octave_linear'):
        e = g['freqlist_' + kind]
        fl, nb, fmax = pkg.get_freqlist(e['args'][0], e['args'][1], kind, e['args'][2])
        assert [float(x) for x in fl] == e['freqlist'], kind
        assert nb == e['nbands'] and float(fmax) == e['fmax']
    assert pkg.get_winlenlist('adaptive', 8, 50, 60, 30) == g['winlen_adaptive'] == [60, 55, 51, 47, 42, 38, 34, 30]
    assert pkg.get_winlenlist('constant', 5, 50, 60, 30) == g['winlen_constant']
    fl, _, _ = pkg.get_freqlist(0.1, 5.0, 'log', 8)
    assert fl[0] == 0.10000000000000005            # math.log(0.1, 10) quirk of helpers.py:30
    rij = pkg.get_rij(g['get_rij']['lat'], g['get_rij']['lon'], 4)
    np.testing.assert_allclose(rij, np.array(g['get_rij']['rij']), rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        pkg.get_rij([1.0, 2.0], [1.0], 2)


def test_make_float_and_txt_roundtrip(tmp_path, capsys):
    assert pkg.make_float([np.float32(1.5), 2]).dtype == np.float64
    vel = np.array([[0.3, 0.31, 0.0], [0.4, 0.41, 0.42]])
    baz = vel * 100
    md = vel / 2
    t = np.array([[1.0, 2.0, 0.0], [1.5, 2.5, 3.5]])
    pkg.write_txtfile(str(tmp_path) + '/', 'res', vel, baz, md, t, [0.5, 1.0, 2.0], [2, 3])
    capsys.readouterr()
    head = open(str(tmp_path) + '/res.txt').readline()
    assert head == 'Fmin \t Fmax \t Time \t Trace_vel \t Backaz \t MdCCM \n'
    v2, b2, m2, t2, fl, ncl, nb, fmin, fmax = pkg.read_txtfile(str(tmp_path) + '/', 'res')
    assert nb == 2 and list(ncl) == [2, 3] and list(fl) == [0.5, 1.0, 2.0] and (fmin, fmax) == (0.5, 2.0)
    np.testing.assert_array_equal(v2[1], vel[1])
    np.testing.assert_array_equal(v2[0, :2], vel[0, :2])


def test_stream_duck_type():
    st = synthetic.make_stream(np.arange(12.0).reshape(3, 4), 2.0, starttime=100.0)
    c = st.copy()
    c[0].data = c[0].data * 2
    assert st[0].data[1] == 1.0 and c[0].data[1] == 2.0 and len(st) == 3
    np.testing.assert_array_equal(st[1].times('matplotlib'), 100.0 + (np.arange(4) / 2.0) / 86400.0)
    assert np.asarray(st[2]).shape == (4,) and st[2].stats.npts == 4
    assert start_datenum(np.datetime64('1970-01-02T00:00:00')) == 1.0


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'nbls.h')).read()
    declared = sorted(set(re.findall(r'\b(nbls_[a-z0-9_]+)\s*\(', header)))
    assert len(declared) >= 15
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_hip.EXPORTS) == declared
    assert _hip.load_library().nbls_version() >= 100


def test_fails_loudly_without_gpu_or_library(tmp_path):
    with pytest.raises(ImportError):
        _hip.load_library(str(tmp_path / 'nope.so'))
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_hip.NblsError):
        _hip.Handle(0)
    c = synthetic.build_config('cfg1', 0.05)
    with pytest.raises(_hip.NblsError):
        pkg.ltsva(c['st'], None, None, 5.0, 0.5, 1.0, rij=c['rij'])


def test_package_does_not_import_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, 'narrow_band_least_squares_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(root, f)).read()
                assert 'nbls_oracle' not in txt.replace('oracle/nbls_oracle.py', ''), f


def test_lts_plan_matches_oracle_constants(oracle):
    rng = np.random.default_rng(1)
    for n, alpha in ((6, 0.75), (8, 0.5), (16, 0.5)):
        rij = rng.uniform(-1, 1, size=(2, n))
        xij, idx, xpinv = planner.co_array(rij)
        xo, io = oracle.co_array(rij)
        np.testing.assert_array_equal(xij, xo)
        assert [tuple(r) for r in idx] == io
        lp = planner.lts_plan(xij, alpha)
        h, raw, rew = oracle.lts_scale_tables(xij.shape[0], alpha)
        assert lp['h'] == h and lp['raw_factor'] == raw
        np.testing.assert_array_equal(lp['rew_table'], rew)
        xs = xij / (oracle.MAD_CONST * np.median(np.abs(xij), axis=0))
        np.testing.assert_array_equal(lp['starts'], oracle.lts_starts(xs))
        assert lp['quantile'] == oracle.LTS_QUANTILE and lp['csteps'] == oracle.LTS_CSTEPS
    W, inc, nwin = planner.window_plan(24001, 20.0, 30, 0.5)
    assert (W, inc, nwin) == (600, 300, 79)
    with pytest.raises(RuntimeError):
        planner.co_array(np.vstack((np.arange(4.0), np.zeros(4))))
    with pytest.raises(ValueError):
        planner.design_bandpass('bessel', 1, 2, 2, 0.1, 20.0)


def test_shard_bands_is_a_balanced_partition():
    costs = dist.band_costs(864000, 40.0, [60, 55, 51, 47, 42, 38, 34, 30] * 6, 0.5, 28)
    for world in (1, 2, 4, 8):
        shards = dist.shard_bands(costs, world)
        assert sorted(b for s in shards for b in s) == list(range(48))
        loads = [sum(costs[b] for b in s) for s in shards]
        assert max(loads) <= 1.2 * (sum(costs) / world)
    assert dist.shard_bands([1.0, 1.0], 4) == [[0], [1], [], []]
    assert dist.dist_info()[1] == 1


@pytest.mark.parametrize('gold,mode', [('loop_ols_butter_linear', 'bands'), ('loop_lts_butter_octave', 'bands'),
                                       ('loop_ols_butter_linear', 'windows'), ('loop_lts_butter_octave', 'windows')])
def test_band_sharded_path_world_size_2_gloo(gold, mode):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29533',
           os.path.join(ROOT, 'tests', '_dist_worker.py'), gold, mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'DIST_OK world=2' in r.stdout


def test_confidence_intervals_against_brute_force(oracle):
    """Product (closed-form tangents + Newton-refined radial extrema) vs the oracle's dense sampling of
    the Szuberla & Olson confidence ellipse; origin-inside and exact-fit cases included."""
    from narrow_band_least_squares_amd.uncertainty import confidence_intervals
    rng = np.random.default_rng(12)
    xij, _ = oracle.co_array(rng.uniform(-1, 1, size=(2, 7)))
    z = rng.standard_normal((2, 12)) * 2.5
    sig = np.abs(rng.standard_normal(12)) * 0.05
    sig[3] = 0.0                      # exact fit
    sig[5] = 50.0                     # huge ellipse: the origin is inside
    z[:, 7] = np.nan
    civ, cib = confidence_intervals(xij, z.T, sig)
    ov, ob = oracle.confidence_intervals(xij, z, sig)
    np.testing.assert_allclose(civ, ov, rtol=1e-7, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(cib, ob, rtol=1e-7, atol=1e-9, equal_nan=True)
    assert civ[3] == 0.0 and cib[3] == 0.0 and np.isnan(cib[5]) and np.isfinite(civ[5])
    assert np.isnan(civ[7]) and np.isnan(cib[7])
    # a tight ellipse far from the origin: half-widths follow the small-angle formulas
    zz = np.array([[3.0], [0.0]])
    ev = np.linalg.eigvalsh(xij.T @ xij)
    cv, cb = confidence_intervals(xij, zz.T, np.array([1e-6]))
    assert cb[0] < 1e-3 and cv[0] < 1e-6


def test_stdict_packing_matches_oracle(oracle):
    """Vectorised dropped-element dictionary (engine.stdict_from_weights) against the oracle's loop:
    same keys (str of the window time), same element lists, band prefix as the reference builds it."""
    from narrow_band_least_squares_amd import engine
    from narrow_band_least_squares_amd.narrow_band_least_squares import _band_prefix, _prefix_stdict
    rng = np.random.default_rng(5)
    xij, pair_idx, _ = planner.co_array(rng.standard_normal((2, 8)))
    io = oracle.co_array(rng.standard_normal((2, 8)))[1]
    assert np.array_equal(np.asarray(io), pair_idx)
    for frac, nwin in ((0.0, 50), (0.07, 333), (0.6, 40), (1.0, 7)):
        wts = (rng.random((nwin + 5, 28)) >= frac).astype(np.uint8)
        t = 17884.0729166667 + np.arange(nwin + 5) * (15.0 / 86400) + rng.random(nwin + 5) * 1e-9
        exp = oracle.stdict_from_weights(wts[:nwin].T, io, t[:nwin], 8)
        got = engine.stdict_from_weights(wts, nwin, t, pair_idx, 8)
        assert got.keys() == exp.keys() and got['size'] == 8
        for k in exp:
            if k != 'size':
                np.testing.assert_array_equal(got[k], exp[k])
        pre = engine.stdict_from_weights(wts, nwin, t, pair_idx, 8, prefix=_band_prefix(7))
        ref = _prefix_stdict(exp, 7)
        assert pre.keys() == ref.keys()


def test_hot_kernels_use_no_scratch(tmp_path):
    """Register spills of the per-workgroup kernels turn into HBM traffic (an unrolled variant of the
    screening kernel once wrote 1 GB of scratch per launch): the kernels of the cfg-3 path must compile
    with private_segment_fixed_size == 0."""
    hipcc = '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    csrc = os.path.join(ROOT, 'narrow_band_least_squares_amd', 'csrc')
    want = {'xcorr_screen.hip': (['screen_kernel', 'quantize_reg_kernelILi4E', 'verify_lds_kernel'], []),
            'solve.hip': (['solve_lts_wave_kernelILi28E', 'solve_ols_kernel'], ['-ffp-contract=off'])}
    for src, (kernels, flags) in want.items():
        out = tmp_path / (src + '.s')
        subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + csrc, '-S', '--cuda-device-only',
                        os.path.join(csrc, src), '-o', str(out)] + flags, check=True, stderr=subprocess.DEVNULL, timeout=600)
        txt = out.read_text()
        for k in kernels:
            m = re.search(r'\.amdhsa_kernel \S*' + k + r'.*?\.amdhsa_private_segment_fixed_size (\d+)', txt, re.S)
            assert m, k
            assert int(m.group(1)) == 0, '%s spills %s bytes per lane' % (k, m.group(1))
