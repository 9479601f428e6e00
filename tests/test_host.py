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


def test_transport_library_is_named_by_a_call_not_by_the_environment(monkeypatch):
    """the shipped library read two test hooks from the environment (NBLS_RCCL_LIB, NBLS_ALLOW_SHARED_DEVICE).
    Now the stand-in transport of the one-GPU rehearsals is named with nbls_comm_set_library: a path that does not
    exist makes the next comm call fail with NBLS_ERR_COMM (nothing else is tried), the environment variable is
    ignored, and NULL restores RCCL by its usual names."""
    import ctypes as C
    from narrow_band_least_squares_amd import _hip
    lib = _hip.load_library()
    monkeypatch.setenv('NBLS_RCCL_LIB', '/nonexistent/from_the_environment.so')
    assert lib.nbls_comm_set_library(b'/nonexistent/libnot_rccl.so', 1) == 0
    buf = (C.c_char * 128)()
    assert lib.nbls_comm_unique_id(buf, 128) == _hip.NBLS_ERR_COMM
    assert lib.nbls_comm_set_library(None, 0) == 0
    src = open(os.path.join(ROOT, 'narrow_band_least_squares_amd', 'csrc', 'comm.hip')).read()
    assert 'getenv' not in src and 'getenv' not in open(os.path.join(ROOT, 'narrow_band_least_squares_amd', 'csrc', 'api.hip')).read()


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
    assert dist.env_rank()[1] == 1
    # contiguous shares (rank order = band order: the one-process form streams the dictionary rank by rank) wherever they
    # balance within 5 % of the LPT partition
    for world in (1, 2, 3, 4, 8):
        shards, contiguous = dist.plan_shards(costs, world)
        assert sorted(b for s in shards for b in s) == list(range(48))
        if contiguous:
            assert all(s == list(range(s[0], s[-1] + 1)) for s in shards if s) and [s[0] for s in shards if s] == sorted(s[0] for s in shards if s)
            assert max(sum(costs[b] for b in s) for s in shards) <= 1.05 * max(sum(costs[b] for b in s) for s in dist.shard_bands(costs, world))
    assert dist.plan_shards([1.0] * 48, 8) == ([list(range(6 * r, 6 * r + 6)) for r in range(8)], True)
    assert dist.shard_bands_contiguous([5, 1, 1, 1], 3) == [[0], [1], [2, 3]]


@pytest.mark.parametrize('gold,mode,world', [('loop_ols_butter_linear', 'bands', 3), ('loop_lts_butter_octave', 'bands', 2),
                                             ('loop_lts_2octave', 'windows', 4), ('loop_ols_cheby1_adaptive', 'bands', 8)])
def test_sharded_path_one_process_many_devices(gold, mode, world, monkeypatch):
    """Host logic of narrow_band_least_squares_parallel() in its one-process form (a handle per device,
    per-device launch threads, gather to root 0) with oracle-backed stand-ins for the device pass."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import _dist_worker
    monkeypatch.setenv('NBLS_STREAM_RESULTS', '1')        # (goldens are small: left to itself a call of this size is not streamed)
    assert _dist_worker.run_single_process(gold, mode, world, monkeypatch) >= 1


@pytest.mark.parametrize('gold,mode,world', [('loop_lts_butter_octave', 'bands', 2), ('loop_lts_2octave', 'windows', 2)])
def test_sharded_path_runs_a_share_in_hbm_rounds(gold, mode, world, monkeypatch):
    """A rank whose share of the bands exceeds the HBM budget of one pass (the sharded call had lost the
    rounds of engine.process) runs it in consecutive passes, assembles its block on the host and loads it back for
    the ONE gather: same tuple as the golden call."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import _dist_worker
    assert _dist_worker.run_single_process(gold, mode, world, monkeypatch, bands_per_pass=1) >= 2


@pytest.mark.parametrize('gold,mode', [('loop_ols_butter_linear', 'bands'), ('loop_lts_butter_octave', 'bands'),
                                       ('loop_ols_butter_linear', 'windows'), ('loop_lts_butter_octave', 'windows'),
                                       ('loop_lts_butter_octave', 'fail'), ('loop_lts_butter_octave', 'fail_early')])
def test_band_sharded_path_world_size_2_gloo(gold, mode):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', NBLS_STREAM_RESULTS='1')    # (streamed although the goldens are small)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29533',
           os.path.join(ROOT, 'tests', '_dist_worker.py'), gold, mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'DIST_OK world=2' in r.stdout


def test_trace_identity_and_row_pipeline_threshold(monkeypatch):
    """Host logic of the two upload shortcuts: a resident trace is recognised by WHERE its samples lie (same buffers, same
    sampling rate — a copy, a float32 or strided row is not "the same trace"), and a pass is queued on rows that are still
    going up only for traces whose upload is worth hiding."""
    from narrow_band_least_squares_amd import engine
    rows = [np.zeros(1000) for _ in range(4)]
    k = engine._trace_key(rows, 20.0)
    assert k is not None and k == engine._trace_key(list(rows), 20.0)
    assert k != engine._trace_key(rows, 40.0)
    assert k != engine._trace_key([r.copy() for r in rows], 20.0)
    assert k != engine._trace_key(rows[:3], 20.0)
    assert engine._trace_key([r.astype(np.float32) for r in rows], 20.0) is None
    assert engine._trace_key([np.zeros(2000)[::2] for _ in range(4)], 20.0) is None
    block = np.zeros((4, 1000))
    assert engine._trace_key(block, 20.0) == engine._trace_key(block, 20.0) != engine._trace_key(block.copy(), 20.0)
    assert not engine.row_pipeline_for(8, 864000)                 # cfg-3: 55 MB
    assert engine.row_pipeline_for(16, 8640000)                   # cfg-4: 1.1 GB
    monkeypatch.setattr(engine, 'ROW_PIPELINE', False)
    assert not engine.row_pipeline_for(16, 8640000)
    monkeypatch.setattr(engine, 'ROW_PIPELINE', True)
    monkeypatch.setattr(engine, 'ROW_PIPELINE_MIN_BYTES', 0)
    assert engine.row_pipeline_for(3, 100)


def test_confidence_intervals_against_brute_force(oracle):
    """The stationary-point / tangent form of the confidence intervals — what the GPU's uncertainty_kernel computes,
    restated in the oracle as confidence_intervals_closed_form — against the INDEPENDENT evaluation: the oracle's dense
    sampling of the Szuberla & Olson confidence ellipse; origin-inside and exact-fit cases included.  (The GPU kernel
    itself is compared with the closed form at 1e-9 in tests/test_gpu_parity.py: test_ltsva_parity.)"""
    confidence_intervals = oracle.confidence_intervals_closed_form
    rng = np.random.default_rng(12)
    xij, _ = oracle.co_array(rng.uniform(-1, 1, size=(2, 7)))
    z = rng.standard_normal((2, 12)) * 2.5
    sig = np.abs(rng.standard_normal(12)) * 0.05
    sig[3] = 0.0                      # exact fit
    sig[5] = 50.0                     # huge ellipse: the origin is inside
    z[:, 7] = np.nan
    civ, cib = confidence_intervals(xij, z, sig)
    ov, ob = oracle.confidence_intervals(xij, z, sig)
    np.testing.assert_allclose(civ, ov, rtol=1e-7, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(cib, ob, rtol=1e-7, atol=1e-9, equal_nan=True)
    assert civ[3] == 0.0 and cib[3] == 0.0 and np.isnan(cib[5]) and np.isfinite(civ[5])
    assert np.isnan(civ[7]) and np.isnan(cib[7])
    # a tight ellipse far from the origin: half-widths follow the small-angle formulas
    zz = np.array([[3.0], [0.0]])
    cv, cb = confidence_intervals(xij, zz, np.array([1e-6]))
    assert cb[0] < 1e-3 and cv[0] < 1e-6
    # the frame handed to the GPU (planner.uncertainty_frame) is the oracle's
    ev, evec = np.linalg.eigh(xij.T @ xij)
    fr = planner.uncertainty_frame(xij)
    ang = np.arccos(evec[0, 0])
    np.testing.assert_array_equal(fr, [ev[0], ev[1], np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)])


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
    want = {'xcorr_screen.hip': (['screen_kernel', 'quantize_reg_kernelILi4E', 'verify_lds_kernel', 'verify_dma_kernel'], []),
            'solve.hip': (['solve_lts_wave_kernelILi28E', 'solve_ols_kernel', 'solve_lts_bucket_kernelILi4ELb1E',
                          'solve_lts_bucket_kernelILi2ELb0E'], ['-ffp-contract=off'])}
    for src, (kernels, flags) in want.items():
        out = tmp_path / (src + '.s')
        subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + csrc, '-S', '--cuda-device-only',
                        os.path.join(csrc, src), '-o', str(out)] + flags, check=True, stderr=subprocess.DEVNULL, timeout=600)
        txt = out.read_text()
        for k in kernels:
            m = re.search(r'\.amdhsa_kernel \S*' + k + r'.*?\.amdhsa_private_segment_fixed_size (\d+)', txt, re.S)
            assert m, k
            assert int(m.group(1)) == 0, '%s spills %s bytes per lane' % (k, m.group(1))


def test_bucket_selection_state_machine(tmp_path):
    """csrc/lts_bucket.h — the per-lane order-statistic state machine of the large-array FAST-LTS kernel (the h-th
    smallest |r| by bucket refinement, one start per lane) — compiled for the host and checked against a sort on
    240 000 selections: hostile key sets (exact zeros, heavy ties, keys that differ in their last bits, NaN / Inf,
    denormals, the whole exponent range), any guess of the threshold (incl. the kernel's sampled one), gather passes
    taken at once or after forced extra histogram passes, and the high-word bin function against the 64-bit one
    wherever the kernel is allowed to use it."""
    exe = tmp_path / 'bucket_select_test'
    subprocess.run(['g++', '-O2', '-std=c++17', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'narrow_band_least_squares_amd', 'csrc'),
                    os.path.join(ROOT, 'tests', 'c_caller', 'bucket_select_test.cpp'), '-o', str(exe)], check=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith('ok 240000 ')


def test_sequential_rounds_build_the_dictionary_after_the_skeleton(monkeypatch):
    """More bands than the HBM budget of one pass, ALPHA < 1 (engine.py): the rounds are collected inside
    the launch loop, before the result skeleton / key text exist; their dictionary entries must wait for
    ``host_overlap`` and then appear in band order.  Stand-in handle (no GPU): every window of band b drops pair b."""
    from narrow_band_least_squares_amd import engine, synthetic
    from narrow_band_least_squares_amd.narrow_band_least_squares import narrow_band_least_squares
    c = synthetic.build_config('cfg2', 0.1)
    nb = 5

    class StubHandle:
        def set_trace_shape(self, *a): pass
        def upload_rows(self, rows): pass

    hstub = StubHandle()

    def fake_launch(h, data, prep, **kw):
        before = kw.get('before_execute')
        if before is not None:
            before()
        P = prep.npairs
        mask = np.full((prep.nbands, prep.vector_len, prep.mask_bytes), 0xff, dtype=np.uint8)
        for n in range(prep.nbands):
            b = fake_launch.next_band + n
            mask[n, :, (b % P) >> 3] &= np.uint8(~(1 << ((b % P) & 7)) & 0xff)
        fake_launch.next_band += prep.nbands
        z = np.zeros((prep.nbands, prep.vector_len))
        h.out = dict(vel=z + 1.0, baz=z, mdccm=z, sigma_tau=z, mask=mask)
    fake_launch.next_band = 0
    hstub.fetch_packed = lambda: hstub.out
    monkeypatch.setattr(engine, 'launch', fake_launch)
    monkeypatch.setattr(engine, 'get_handle', lambda device=None, slot=0: hstub)
    data = c['data']
    monkeypatch.setenv('NBLS_MAX_FILTERED_GB', repr(2.5 * 8 * data.size / 2.0 ** 30))          # two bands per pass
    monkeypatch.setenv('NBLS_STREAM_RESULTS', '0')            # (the streamed form of the rounds: next test)
    assert engine.max_bands_per_pass(*data.shape) == 2
    fr = np.logspace(-2, 1, 16)
    w = np.zeros(16)
    out = narrow_band_least_squares(c['WINLEN_list'][:nb], 0.5, 0.5, c['st'], None, None, nb, w, w, c['freqlist'][:nb + 1], 'log',
                                    fr, 'butter', 2, 0.01, rij=c['rij'])
    stdict = out[4]
    keys = [k for k in stdict if k != 'size']
    assert [k[:2] for k in keys] == sorted(k[:2] for k in keys)                 # band order
    assert len(keys) == sum(out[6]) and out[7].shape == (nb, 16)
    pair_idx = np.array([(i, j) for i in range(6) for j in range(i + 1, 6)])
    for b in range(nb):
        first = next(k for k in keys if k.startswith('%02d_' % (b + 1)))
        np.testing.assert_array_equal(np.sort(stdict[first]), np.sort(pair_idx[b % 15] + 1))


def test_streamed_pass_builds_rows_and_dictionary_batch_by_batch(monkeypatch):
    """The default whole call since round 4: ONE pass whose unit batches arrive one after the other (engine.process with
    ``nbls_stream_results``).  Stand-in handle (no GPU) that hands out batches cutting through bands: the rows and the
    dictionary must equal those of the unstreamed call, entries in band order, 'size' behind the first band."""
    from narrow_band_least_squares_amd import engine, synthetic
    from narrow_band_least_squares_amd.narrow_band_least_squares import narrow_band_least_squares
    c = synthetic.build_config('cfg2', 0.1)
    nb = 5
    rng = np.random.default_rng(11)
    monkeypatch.setenv('NBLS_STREAM_RESULTS', '1')        # (a call of this size is not streamed by itself: engine.stream_pays)

    class StubHandle:
        def set_trace_shape(self, *a): pass
        def upload_rows(self, rows): pass

    hstub = StubHandle()
    waited = []

    def fake_launch(h, data, prep, **kw):
        before = kw.get('before_execute')
        if before is not None:
            before()
        h.streamed = bool(kw.get('stream'))
        B, VL, MB, P = prep.nbands, prep.vector_len, prep.mask_bytes, prep.npairs
        w = (rng.random((B, VL, P)) > 0.1).astype(np.uint8)
        mask = np.packbits(w, axis=-1, bitorder='little')
        g = rng.random((4, B, VL))
        for b in range(B):
            g[:, b, prep.nwin[b]:] = 0.0
            mask[b, prep.nwin[b]:] = 0
        h.out = dict(vel=g[0], baz=g[1], mdccm=g[2], sigma_tau=g[3], mask=mask)
        h.nwin, h.VL = [int(x) for x in prep.nwin], VL
    hstub.fetch_packed = lambda: hstub.out

    def batches():
        U = sum(hstub.nwin)
        cuts = sorted({0, U} | {int(x) for x in (0.13 * U, 0.5 * U, 0.51 * U, 0.9 * U)})
        return list(zip(cuts[:-1], cuts[1:]))

    def cell(u):
        off = 0
        for b, n in enumerate(hstub.nwin):
            if u < off + n:
                return b * hstub.VL + (u - off)
            off += n
        raise AssertionError

    hstub.result_batches = lambda: len(batches())

    def wait_result_batch(k):
        u0, u1 = batches()[k]
        waited.append(k)
        c0, c1 = cell(u0), cell(u1 - 1) + 1
        o = hstub.out
        gsrc = np.full((4, len(hstub.nwin) * hstub.VL), np.nan)           # cells of other batches are undefined
        msrc = np.full((len(hstub.nwin) * hstub.VL, o['mask'].shape[2]), 0x55, dtype=np.uint8)
        for i, name in enumerate(('vel', 'baz', 'mdccm', 'sigma_tau')):
            gsrc[i, c0:c1] = o[name].reshape(-1)[c0:c1]
        msrc[c0:c1] = o['mask'].reshape(-1, o['mask'].shape[2])[c0:c1]
        return u0, u1, c0, c1, gsrc, msrc
    hstub.wait_result_batch = wait_result_batch
    monkeypatch.setattr(engine, 'launch', fake_launch)
    monkeypatch.setattr(engine, 'get_handle', lambda device=None, slot=0: hstub)
    monkeypatch.delenv('NBLS_PIPELINE_GROUPS', raising=False)
    fr = np.logspace(-2, 1, 16)
    w = np.zeros(16)
    args = (c['WINLEN_list'][:nb], 0.5, 0.5, c['st'], None, None, nb, w, w, c['freqlist'][:nb + 1], 'log', fr, 'butter', 2, 0.01)
    out = narrow_band_least_squares(*args, rij=c['rij'])
    assert hstub.streamed and waited == list(range(len(batches())))
    o = hstub.out
    np.testing.assert_array_equal(out[0], o['vel'])
    np.testing.assert_array_equal(out[1], o['baz'])
    np.testing.assert_array_equal(out[2], o['mdccm'])
    keys = engine.time_keys(out[3], out[6], ['%02d_' % (b + 1) for b in range(nb)])
    exp = engine._py_stdict_from_mask(o['mask'], np.array(out[6]), planner.pair_table(6), 6, keys)
    assert list(out[4].keys()) == list(exp.keys())
    for k in exp:
        if k != 'size':
            np.testing.assert_array_equal(out[4][k], exp[k])
    # more bands than the HBM budget of one pass: streamed rounds one after the other, dictionary in band order
    data = c['data']
    monkeypatch.setenv('NBLS_MAX_FILTERED_GB', repr(2.5 * 8 * data.size / 2.0 ** 30))          # two bands per pass
    outs = []
    real_fake = fake_launch

    def recording_launch(h, data, prep, **kw):
        real_fake(h, data, prep, **kw)
        outs.append(h.out)
    monkeypatch.setattr(engine, 'launch', recording_launch)
    del waited[:]
    out3 = narrow_band_least_squares(*args, rij=c['rij'])
    assert len(outs) == 3 and hstub.streamed
    m_all = np.concatenate([x['mask'] for x in outs])
    np.testing.assert_array_equal(out3[0], np.concatenate([x['vel'] for x in outs]))
    exp3 = engine._py_stdict_from_mask(m_all, np.array(out3[6]), planner.pair_table(6), 6, keys)
    assert list(out3[4].keys()) == list(exp3.keys())
    for k in exp3:
        if k != 'size':
            np.testing.assert_array_equal(out3[4][k], exp3[k])
    monkeypatch.delenv('NBLS_MAX_FILTERED_GB')
    monkeypatch.setattr(engine, 'launch', fake_launch)
    # the same through the band-group path (NBLS_STREAM_RESULTS=0): not streamed, same kind of result
    monkeypatch.setenv('NBLS_STREAM_RESULTS', '0')
    monkeypatch.setenv('NBLS_PIPELINE_GROUPS', '1')
    del waited[:]
    out2 = narrow_band_least_squares(*args, rij=c['rij'])
    assert not hstub.streamed and not waited
    np.testing.assert_array_equal(out2[0], hstub.out['vel'])
    # left to itself (engine.stream_pays): a call of 10^3 units is fetched in one piece, LTS or not; streaming starts where
    # the dictionary work of a batch is worth overlapping
    monkeypatch.delenv('NBLS_STREAM_RESULTS')
    monkeypatch.delenv('NBLS_PIPELINE_GROUPS')
    del waited[:]
    narrow_band_least_squares(*args, rij=c['rij'])
    assert not hstub.streamed and not waited
    assert engine.stream_pays(0.5, [40000]) and engine.stream_pays(0.5, [1000], npairs=496) and not engine.stream_pays(1.0, [40000])
    assert not engine.stream_pays(0.5, [15999]) and engine.stream_pays(0.5, [8000, 8000])
    monkeypatch.setenv('NBLS_STREAM_RESULTS', '1')
    assert engine.stream_pays(1.0, [10])
    monkeypatch.setenv('NBLS_STREAM_RESULTS', '0')
    assert not engine.stream_pays(0.5, [40000])


def test_bench_refuses_what_it_cannot_measure():
    """`python bench.py --gpus N` without a launcher uses the one-process form on GPUs 0..N-1; with fewer GPUs visible
    (here: none) it must print ONE JSON line with value null / status failed and exit non-zero — never a 1-GPU number
    under n_gpus N.  Same for --shard traces without a launcher."""
    import json
    for extra in (['--gpus', '4'], ['--gpus', '2', '--shard', 'traces']):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', 'cfg2', '--steps', '1', '--warmup', '0'] + extra,
                           capture_output=True, text=True, timeout=300, env=dict(os.environ, NBLS_DEVICES=''))
        assert r.returncode == 2, r.stdout + r.stderr
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line['value'] is None and line['status'] == 'failed' and line['n_gpus'] == int(extra[1]) and line['note']


def test_generated_screen_kloop_is_current():
    """csrc/screen_kloop.inc (the hand-scheduled K loop of the screening kernel) is what its generator prints, and
    its schedule keeps the invariants the kernel relies on: every accumulator is written by exactly the products of
    its tile, LDS waits never exceed the requests in flight, and the block ends with the wait states a vector read
    of a matrix-core result needs."""
    gen = os.path.join(ROOT, 'tools', 'gen_screen_kloop.py')
    inc = os.path.join(ROOT, 'narrow_band_least_squares_amd', 'csrc', 'screen_kloop.inc')
    out = subprocess.run([sys.executable, gen], check=True, capture_output=True, text=True, timeout=60).stdout
    assert out == open(inc).read()
    macros = out.split('#define ')[1:]
    assert [m.split('(')[0] for m in macros] == ['NBLS_SCREEN_KLOOP_ASM', 'NBLS_SCREEN_KLOOP_S1_ASM']
    # (products, accumulator / A fragment / partner fragment register ranges) of the two-block and the one-block loop
    want = [(42, (96, 124), (64, 76), (80, 92)), (84, (192, 252), (160, 188), (144, 156))]
    for text, (nprod, racc, ra, rb) in zip(macros, want):
        body = [ln.split('"')[1].replace('\\n\\t', '') for ln in text.splitlines() if ln.strip().startswith('"')]
        mf = [ln for ln in body if ln.startswith('v_mfma')]
        assert len(mf) == nprod
        prev_dst = None
        for ln in mf:                  # in-place accumulation (or a literal zero as the first addend), never back to back
            dst, a, b, c = [x.strip() for x in ln.split(' ', 1)[1].split(',')]
            assert c in (dst, '0')
            assert dst != prev_dst or c == '0'
            prev_dst = dst
            first = lambda r: int(r[2:].split(':')[0])
            assert racc[0] <= first(dst) <= racc[1] and ra[0] <= first(a) <= ra[1] and rb[0] <= first(b) <= rb[1]
        assert body[-2:] == ['s_nop 15', 's_nop 7']
        # LDS requests in flight never exceed what the 4-bit lgkmcnt can count, on every path through the block
        inflight, at_branch = 0, {}
        for ln in body:
            if ln.startswith('ds_read'):
                inflight += 1
            elif ln.startswith('s_waitcnt lgkmcnt('):
                inflight = min(inflight, int(ln.split('(')[1].rstrip(')')))
            elif ln.startswith('s_cbranch_scc1 ') or ln.startswith('s_branch '):
                tgt = ln.split()[1][0]
                at_branch[tgt] = max(at_branch.get(tgt, 0), inflight)
            elif ln in ('1:', '2:', '3:', '4:'):
                inflight = max(inflight, at_branch.get(ln[0], 0))
            assert inflight <= 15, ln


def test_host_extension_matches_python_equivalents():
    """csrc/host_ext.cpp (_nbls_host): float repr identical to Python's for every kind of value, the key
    list and the dictionary identical (keys, order, values, dtype, read-only flag) to the NumPy forms."""
    from narrow_band_least_squares_amd import engine
    ext = engine._hostext
    assert ext is not None, 'narrow_band_least_squares_amd/_nbls_host.so is not built (make -C .../csrc)'
    import random
    import struct
    rng = random.Random(3)
    vals = [0.0, -0.0, 1.0, 1e16, 1e15, 9999999999999998.0, 1.5e16, 1e-4, 1e-5, 0.00012345, 123456789012345680.0,
            float('inf'), float('-inf'), 5e-324, 1.7976931348623157e308, 17884.072916666701, 100.0, 1e22, 1e23, 0.1]
    for _ in range(60000):
        k = rng.random()
        if k < 0.3:
            x = struct.unpack('d', struct.pack('Q', rng.getrandbits(64)))[0]
        elif k < 0.7:
            x = 17884.0 + rng.random() * 400
        else:
            x = rng.uniform(-1, 1) * 10 ** rng.randint(-20, 20)
        if x == x:
            vals.append(x)
    for x in vals:
        assert ext.float_repr(x) == repr(x) == str(np.float64(x))
    assert ext.float_repr(float('nan')) == 'nan'
    r = np.random.default_rng(4)
    for P, nch in ((28, 8), (15, 6), (120, 16), (3, 3)):
        pair_idx = planner.pair_table(nch)
        B, VL = 5, 37
        nwin = np.array([37, 30, 0, 36, 1])
        for frac in (0.0, 0.02, 0.5):
            w = (r.random((B, VL, P)) >= frac).astype(np.uint8)
            w[1, 3:20] = 1
            w[1, 3:20, :2] = 0                      # a run of equal patterns
            mask = np.packbits(w, axis=-1, bitorder='little')
            t = 17884.07 + r.random((B, VL))
            pre = ['%s_' % str(b + 98).zfill(2) for b in range(B)]          # '98_', '99_', '100_', ...
            keys = ext.time_keys(t, nwin, pre)
            assert keys == engine._py_time_keys(t, nwin, pre) and len(keys) == nwin.sum()
            assert ext.time_keys(t, nwin, None) == engine._py_time_keys(t, nwin, None)
            got = ext.build_stdict(mask, nwin, pair_idx, nch, keys)
            exp = engine._py_stdict_from_mask(mask, nwin, pair_idx, nch, keys)
            assert list(got.keys()) == list(exp.keys()) and got['size'] == nch
            for k in exp:
                if k != 'size':
                    np.testing.assert_array_equal(got[k], exp[k])
                    assert got[k].dtype == exp[k].dtype and not got[k].flags.writeable
            # built group by group (pipelined call): same dictionary, same order
            for impl in (ext.build_stdict, engine._py_stdict_from_mask):
                inc = {}
                for b0, b1 in ((0, 2), (2, 3), (3, 5)):
                    ret = impl(mask[b0:b1], nwin[b0:b1], pair_idx, nch, keys, inc, int(nwin[:b0].sum()))
                    assert ret is inc
                assert list(inc.keys()) == list(exp.keys())
                for k in exp:
                    if k != 'size':
                        np.testing.assert_array_equal(inc[k], exp[k])
            # built unit batch by unit batch (streamed pass): ranges that start and end inside bands, one of them empty, some
            # of them inside the first band — same dictionary, same order, 'size' behind the first band's entries
            tot = int(nwin.sum())
            for cuts in ((0, 5, 36, 37, 38, 70, 70, tot), (0, tot), (0, 37, tot), (0, 40, tot)):
                for impl in (ext.build_stdict, engine._py_stdict_from_mask):
                    inc = {}
                    for u0, u1 in zip(cuts[:-1], cuts[1:]):
                        if impl is ext.build_stdict:
                            ret = impl(mask, nwin, pair_idx, nch, keys, inc, 0, None, u0, u1)
                        else:
                            ret = impl(mask, nwin, pair_idx, nch, keys, inc, 0, (u0, u1))
                        assert ret is inc
                    assert list(inc.keys()) == list(exp.keys()), (cuts, impl)
                    for k in exp:
                        if k != 'size':
                            np.testing.assert_array_equal(inc[k], exp[k])
            # value arrays shared across the groups' calls through a caller-held cache: same dictionary, and a pattern
            # that shows up in two groups is ONE object
            cache = ext.new_pattern_cache()
            inc = {}
            for b0, b1 in ((0, 2), (2, 3), (3, 5)):
                ext.build_stdict(mask[b0:b1], nwin[b0:b1], pair_idx, nch, keys, inc, int(nwin[:b0].sum()), cache)
            assert list(inc.keys()) == list(exp.keys())
            by_bytes = {}
            for k in exp:
                if k != 'size':
                    np.testing.assert_array_equal(inc[k], exp[k])
                    assert by_bytes.setdefault(inc[k].tobytes(), inc[k]) is inc[k]
            # the cache follows the pair table: another array size resets it instead of handing out stale arrays
            other = planner.pair_table(4)
            m4 = np.packbits((r.random((1, 5, len(other))) > 0.3).astype(np.uint8), axis=-1, bitorder='little')
            k4 = ext.time_keys(t[:1, :5], np.array([5]), None)
            d4 = ext.build_stdict(m4, np.array([5]), other, 4, k4, None, 0, cache)
            e4 = engine._py_stdict_from_mask(m4, np.array([5]), other, 4, k4)
            assert list(d4) == list(e4) and all(np.array_equal(d4[k], e4[k]) for k in e4)
            with pytest.raises(TypeError):
                ext.build_stdict(mask, nwin, pair_idx, nch, keys, None, 0, object())
            # key TEXT instead of key objects (formatted on 1 and 3 threads): the same text, the same dictionary
            for nt in (1, 3):
                text, length = ext.time_key_text(t, nwin, pre, nt)
                assert text.shape == (len(keys), 40) and length.shape == (len(keys),)
                assert [bytes(text[i, :length[i]]).decode('ascii') for i in range(len(keys))] == keys
                for impl in (ext.build_stdict, engine._py_stdict_from_mask):
                    inc = {}
                    for b0, b1 in ((0, 2), (2, 3), (3, 5)):
                        impl(mask[b0:b1], nwin[b0:b1], pair_idx, nch, (text, length), inc, int(nwin[:b0].sum()))
                    assert list(inc.keys()) == list(exp.keys())
                    assert all(type(k) is str for k in inc)
                    for k in exp:
                        if k != 'size':
                            np.testing.assert_array_equal(inc[k], exp[k])
    with pytest.raises(ValueError):
        ext.build_stdict(np.zeros((1, 2, 4), np.uint8), np.array([2]), planner.pair_table(8), 8, ['a'])
    with pytest.raises(ValueError):                          # a prefix that does not fit the 40-byte text slots
        ext.time_key_text(np.zeros((1, 2)), np.array([2]), ['x' * 20], 1)
    with pytest.raises(ValueError):                          # malformed key text
        ext.build_stdict(np.zeros((1, 2, 4), np.uint8), np.array([2]), planner.pair_table(8), 8,
                         (np.zeros((2, 39), np.uint8), np.zeros(2, np.uint8)))


def test_one_filter_design_serves_obspy_and_scipy_forms():
    """planner._design_cached returns ONE SOS for the band-pass obspy applies (corners normalised by
    fs/2, zpk -> zpk2sos) and the one helpers.py:128 designs for the response plot (fs=Fs): the two are the same
    floating-point numbers."""
    from scipy import signal
    for order in (1, 2, 3, 4, 8):
        for fs in (20.0, 40.0, 100.0, 37.3):
            fl = np.logspace(np.log10(0.05), np.log10(fs * 0.49), 25)
            for i in range(24):
                z, p, k = signal.iirfilter(order, [fl[i] / (0.5 * fs), fl[i + 1] / (0.5 * fs)], btype='band', ftype='butter',
                                           output='zpk')
                a = signal.zpk2sos(z, p, k)
                b = signal.iirfilter(order, [fl[i], fl[i + 1]], btype='band', ftype='butter', fs=fs, output='sos')
                np.testing.assert_array_equal(a, b)
                sa, zp, sr = planner.design_bandpass('butter', fl[i], fl[i + 1], order, 0.01, fs)
                np.testing.assert_array_equal(sa, a)
                np.testing.assert_array_equal(sr, b)
                assert zp is True


def test_txtfile_matches_the_references_own_bytes(tmp_path, capsys):
    """f3: the product's write_txtfile produces byte for byte the file the REFERENCE's
    write_txtfile wrote for the same grids (tests/golden/txtfile_ref.txt, helpers.py:161-182), and its
    read_txtfile returns what the reference's read_txtfile returned for that file (txtfile_read.npz,
    helpers.py:185-235).  Fixtures: tests/golden/make_goldens.py extra."""
    from narrow_band_least_squares_amd import write_txtfile, read_txtfile
    gd = os.path.join(ROOT, 'tests', 'golden')
    g = np.load(os.path.join(gd, 'loop_ols_butter_linear.npz'), allow_pickle=False)
    d = str(tmp_path) + '/'
    write_txtfile(d, 'mine', g['vel'], g['baz'], g['mdccm'], g['t'], g['freqlist'], list(g['num_compute']))
    assert capsys.readouterr().out.split() == [str(n) for n in g['num_compute']]       # the reference prints them
    ref_bytes = open(os.path.join(gd, 'txtfile_ref.txt'), 'rb').read()
    assert open(d + 'mine.txt', 'rb').read() == ref_bytes
    exp = np.load(os.path.join(gd, 'txtfile_read.npz'), allow_pickle=False)
    open(d + 'ref.txt', 'wb').write(ref_bytes)
    out = read_txtfile(d, 'ref')
    names = ('vel', 'baz', 'mdccm', 't', 'freqlist', 'num_compute', 'nbands', 'fmin', 'fmax')
    ncl = np.asarray(out[5])
    np.testing.assert_array_equal(ncl, exp['num_compute'])
    for n, v in zip(names, out):
        v = np.array(v, dtype=float)
        if v.ndim == 2:
            for b in range(v.shape[0]):
                v[b, ncl[b]:] = 0.0                  # np.empty tails, as in the reference
        np.testing.assert_array_equal(v, exp[n], err_msg=n)


# ---- checks of the recalled FAST-LTS / geodesy pieces that do NOT share text with the oracle --------------
# (tests/test_host.py::test_lts_plan_matches_oracle_constants compares planner.py with its restatement in the
#  oracle; the tests below pin the same functions to hand-worked numbers, closed forms and published constants.)

def test_lts_subset_size_table():
    """h = 2*floor((P+p+1)/2) - P + 2*(P - floor((P+p+1)/2))*alpha, p = 2 (robustbase h.alpha.n): worked by hand."""
    table = {(15, 0.75): 12, (28, 0.5): 15, (120, 0.5): 61, (496, 0.5): 249, (6, 0.5): 4, (10, 0.5): 6, (21, 0.5): 12,
             (15, 0.5): 9, (28, 0.75): 21, (28, 0.9): 25, (120, 0.75): 90, (36, 0.5): 19}
    for (P, alpha), h in table.items():
        assert planner.lts_h(P, alpha) == h, (P, alpha)
    for P in range(3, 600):
        assert planner.lts_h(P, 0.5) == (P + 3) // 2           # the smallest subset that still has a majority
        assert planner.lts_h(P, 0.999999) in (P - 1, P)
        hs = [planner.lts_h(P, a) for a in np.linspace(0.5, 0.99, 25)]
        assert all(b >= a for a, b in zip(hs, hs[1:]))          # monotone in alpha


def test_lcg_subsets_by_closed_form():
    """robustbase's uniran: seed <- (seed*5761 + 999) mod 65536 from seed 0.  First seeds worked by hand, the
    whole sequence against the closed form seed_n = 999*(5761^n - 1)/5760 mod 65536 (big integers, no loop), full
    period (Hull-Dobell: c odd, a-1 divisible by 4), and the subsets rebuilt from that sequence."""
    def seed(n):
        return (999 * (5761 ** n - 1) // 5760) % 65536
    assert [seed(n) for n in (1, 2, 3, 4)] == [999, 54606, 13365, 57500]
    assert len({seed(n) for n in range(1, 4097)}) == 4096
    assert seed(65536) == 0 and seed(32768) != 0                 # period exactly 65536
    for P in (120, 496, 36):
        subs = planner.uniran_subsets(P)
        assert subs.shape == (500, 2)
        n, rebuilt = 0, []
        for _ in range(500):
            pick = []
            while len(pick) < 2:
                n += 1
                idx = int(seed(n) / 65536.0 * P)
                if idx not in pick:
                    pick.append(idx)
            rebuilt.append(pick)
        assert subs.tolist() == rebuilt
    assert planner.uniran_subsets(120)[:2].tolist() == [[1, 99], [24, 105]]        # by hand from the four seeds above


def test_consistency_factor_by_numerical_integration():
    """The raw / reweighted consistency factor is 1/sqrt(E[x^2 | |x| <= q]) of a standard normal truncated to the
    central m/n of its mass — checked by quadrature, plus its limits."""
    from scipy import integrate
    from scipy.stats import norm
    for m, n in ((15, 28), (61, 120), (249, 496), (12, 15), (27, 28), (3, 28)):
        q = norm.ppf(0.5 * (1 + m / n))
        second, _ = integrate.quad(lambda x: x * x * norm.pdf(x), -q, q, epsabs=1e-13, epsrel=1e-13)
        expect = 1.0 / np.sqrt(second / (m / n))
        assert abs(planner._consfactor(m, n) - expect) < 1e-10 * expect
        assert planner._consfactor(m, n) > 1.0
    assert planner._consfactor(28, 28) == 1.0
    tab = planner._consfactor_table(200)
    assert np.all(np.diff(tab) < 0) and abs(tab[-1] - planner._consfactor(199, 200)) == 0.0     # -> 1 as m -> n
    assert tab[-1] < 1.15 and tab[0] > 50


def test_small_sample_correction_closed_form():
    """LTScnp2-style finite-sample factor: fp(n) = 1 - exp(k0)/n^k1 with (k0, k1) fitted through the tabulated
    points n = 3p^2 and 5p^2, interpolated in alpha between 0.5 and 0.875 and on to 1 at alpha = 1.  Evaluated here
    through the explicit two-point solution instead of the linear solve."""
    import math
    p = 2

    def fp(c, n):
        y0 = math.log(-c[0][0] / p ** c[1][0])
        y1 = math.log(-c[0][1] / p ** c[1][1])
        k1 = (y0 - y1) / math.log(c[2][1] / c[2][0])
        k0 = y0 + k1 * math.log(c[2][0] * p * p)
        return 1.0 - math.exp(k0) / n ** k1
    for n in (15, 28, 120, 496):
        for alpha in (0.5, 0.6, 0.75, 0.875, 0.9, 0.99):
            for c500, c875 in ((planner._RAW_500, planner._RAW_875), (planner._REW_500, planner._REW_875)):
                f5, f8 = fp(c500, n), fp(c875, n)
                f = f5 + (f8 - f5) / 0.375 * (alpha - 0.5) if alpha <= 0.875 else f8 + (1 - f8) / 0.125 * (alpha - 0.875)
                got = planner._cnp2(p, n, alpha, c500, c875)
                assert abs(got - 1.0 / f) < 1e-12 * abs(got)
    # the correction shrinks with the sample size and vanishes at alpha = 1
    assert planner._cnp2(2, 28, 0.5, planner._RAW_500, planner._RAW_875) > planner._cnp2(2, 496, 0.5, planner._RAW_500, planner._RAW_875) > 1.0
    assert abs(planner._cnp2(2, 28, 1.0, planner._RAW_500, planner._RAW_875) - 1.0) < 1e-12
    lp = planner.lts_plan(planner.co_array(np.random.default_rng(2).standard_normal((2, 8)))[0], 0.5)
    assert abs(lp['quantile'] - 2.241402727604947) < 1e-12       # qnorm(0.9875)


def test_vincenty_against_published_constants():
    """WGS84 quarter meridian 10 001 965.729 m and one degree of equatorial longitude a*pi/180 (published values),
    reciprocity of the azimuths, and agreement with the spherical law of cosines on a short line."""
    from narrow_band_least_squares_amd.helpers import vincenty_inverse
    d, az, _ = vincenty_inverse(0.0, 10.0, 90.0, 10.0)
    assert abs(d - 10001965.729) < 2e-3 and abs(az) < 1e-9
    d, az, baz = vincenty_inverse(0.0, 0.0, 0.0, 1.0)
    assert abs(d - 6378137.0 * np.pi / 180) < 1e-6 and abs(az - 90.0) < 1e-9 and abs(baz - 270.0) < 1e-9
    d12, a12, b12 = vincenty_inverse(64.87, -147.86, 64.875, -147.85)
    d21, a21, b21 = vincenty_inverse(64.875, -147.85, 64.87, -147.86)
    assert abs(d12 - d21) < 1e-6 and abs(a12 - b21) < 1e-7 and abs(b12 - a21) < 1e-7
    # sphere of the local radius of curvature: agrees to 1e-4 relative over ~700 m
    lat = np.radians(64.8725)
    a_, f_ = 6378137.0, 1 / 298.257223563
    e2 = f_ * (2 - f_)
    M = a_ * (1 - e2) / (1 - e2 * np.sin(lat) ** 2) ** 1.5
    Nn = a_ / np.sqrt(1 - e2 * np.sin(lat) ** 2)
    dy = np.radians(0.005) * M
    dx = np.radians(0.01) * Nn * np.cos(lat)
    assert abs(d12 - np.hypot(dx, dy)) < 1e-4 * d12


def test_host_extension_under_address_and_ub_sanitizers(tmp_path):
    """SURVEY.md §5 (sanitizers, host side only): csrc/host_ext.cpp built with -fsanitize=address,undefined and
    driven through its whole surface — key text, dictionary assembly (fresh, incremental, > 8 mask bytes), every
    error path — in a child interpreter with libasan preloaded.  No report, exit code 0."""
    import subprocess
    src = os.path.join(ROOT, 'narrow_band_least_squares_amd', 'csrc', 'host_ext.cpp')
    import sysconfig
    out = str(tmp_path / '_nbls_host.so')
    cmd = ['g++', '-O1', '-g', '-std=c++17', '-fPIC', '-shared', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-pthread',
           '-I' + sysconfig.get_paths()['include'], '-I' + np.get_include(), src, '-o', out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    libasan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import _nbls_host as ext
rng = np.random.default_rng(0)
for x in list(rng.standard_normal(2000) * 10.0 ** rng.integers(-30, 30, 2000)) + [0.0, -0.0, float('inf'), float('nan'), 5e-324, 1.7976931348623157e308]:
    assert ext.float_repr(float(x)) == repr(float(x))
for nch in (3, 8, 16, 32):
    pair = np.array([(i, j) for i in range(nch - 1) for j in range(i + 1, nch)], dtype=np.int32)
    P = len(pair); B, VL = 4, 19
    nwin = np.array([19, 0, 7, 18])
    w = (rng.random((B, VL, P)) > 0.1).astype(np.uint8)
    mask = np.packbits(w, axis=-1, bitorder='little')
    t = 17884.0 + rng.random((B, VL))
    keys = ext.time_keys(t, nwin, ['%%02d_' %% (b + 1) for b in range(B)])
    assert len(keys) == nwin.sum() and len(ext.time_keys(t, nwin, None)) == nwin.sum()
    d = ext.build_stdict(mask, nwin, pair, nch, keys)
    inc = {}
    for b0, b1 in ((0, 1), (1, 3), (3, 4)):
        ext.build_stdict(mask[b0:b1], nwin[b0:b1], pair, nch, keys, inc, int(nwin[:b0].sum()))
    assert list(inc) == list(d) and d['size'] == nch
    cache = ext.new_pattern_cache()
    inc3 = {}
    for b0, b1 in ((0, 1), (1, 3), (3, 4)):
        ext.build_stdict(mask[b0:b1], nwin[b0:b1], pair, nch, keys, inc3, int(nwin[:b0].sum()), cache)
    assert list(inc3) == list(d)
    del cache
    for nt in (1, 3):                    # key text formatted on threads, key objects made on demand
        kt = ext.time_key_text(t, nwin, ['%%02d_' %% (b + 1) for b in range(B)], nt)
        inc2 = {}
        for b0, b1 in ((0, 1), (1, 3), (3, 4)):
            ext.build_stdict(mask[b0:b1], nwin[b0:b1], pair, nch, kt, inc2, int(nwin[:b0].sum()))
        assert list(inc2) == list(d)
    for bad in (lambda: ext.build_stdict(mask, nwin, pair, nch, (kt[0][:3], kt[1][:3])), lambda: ext.build_stdict(mask, nwin, pair, nch, (kt[0], kt[1][:3])),
                lambda: ext.time_key_text(t, nwin, ['y' * 30] * B, 2), lambda: ext.time_key_text(t, nwin + 100, None, 2),
                lambda: ext.build_stdict(mask, nwin[:2], pair, nch, keys), lambda: ext.build_stdict(mask, nwin, pair, nch, keys[:3]),
                lambda: ext.build_stdict(np.concatenate((mask, mask), axis=-1), nwin, pair, nch, keys), lambda: ext.build_stdict(mask, nwin, pair, nch, keys, [], 0),
                lambda: ext.build_stdict(mask, nwin, pair, nch, keys, {}, -1), lambda: ext.time_keys(t, nwin + 100, None),
                lambda: ext.time_keys(t, nwin, ['a']), lambda: ext.time_keys(t[0], nwin, None), lambda: ext.float_repr('x')):
        try:
            bad()
        except (ValueError, TypeError):
            pass
        else:
            raise SystemExit('an invalid call was accepted')
print('SAN_OK')
''' % str(tmp_path)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS='detect_leaks=0', PYTHONMALLOC='malloc')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'SAN_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-4000:]


def test_vectorised_filter_design_is_bitwise_scipy():
    """planner._bandpass_sos_batch restates SciPy's iirfilter(..., output='sos') band-pass chain (prototype ->
    lp2bp_zpk -> bilinear_zpk -> zpk2sos 'nearest') element-wise over many bands: every coefficient, including the
    sign of zeros, equals SciPy's per-band result; band sets it declines (real poles, ties) go to SciPy."""
    from scipy import signal
    checked = 0
    for ftype in ('butter', 'cheby1'):
        for order in (1, 2, 3, 4, 6, 8):
            for fs in (20.0, 40.0, 100.0, 37.3):
                for fl in (np.logspace(np.log10(0.05), np.log10(fs * 0.49), 24), np.linspace(0.3, fs * 0.49, 17),
                           0.05 * 2.0 ** np.arange(0, int(np.log2(fs * 0.49 / 0.05)))):
                    sos = planner._bandpass_sos_batch(ftype, fl[:-1], fl[1:], order, 0.01, fs)
                    if sos is None:
                        continue
                    for i in range(len(fl) - 1):
                        ref = signal.iirfilter(order, [fl[i], fl[i + 1]], rp=0.01, btype='band', ftype=ftype, fs=fs, output='sos')
                        np.testing.assert_array_equal(sos[i], ref)
                        assert not np.any(np.signbit(sos[i]) != np.signbit(ref))
                        checked += 1
    assert checked > 1500
    # the public entry: same triples as design_bandpass, cache shared, Nyquist band handed to the scalar path
    planner.design_cache_clear()
    edges = [(0.1, 0.2), (1.0, 3.0), (4.0, 9.0)]
    many = planner.design_bandpass_many('butter', edges, 2, 0.01, 20.0)
    for (lo, hi), (sa, zp, sr) in zip(edges, many):
        a, z, r = planner.design_bandpass('butter', lo, hi, 2, 0.01, 20.0)
        np.testing.assert_array_equal(sa, a)
        np.testing.assert_array_equal(sr, r)
        assert zp is z is True
    with pytest.warns(UserWarning):
        planner.design_bandpass_many('butter', [(1.0, 9.999995)], 2, 0.01, 20.0)    # within 1e-6 of Nyquist: obspy's high-pass fallback
    with pytest.raises(ValueError):
        planner.design_bandpass_many('butter', [(1.0, 10.0)], 2, 0.01, 20.0)        # at Nyquist helpers.py:128's own design raises
    with pytest.raises(ValueError):
        planner.design_bandpass_many('bessel', [(1.0, 2.0)], 2, 0.01, 20.0)
    assert planner.design_bandpass_many('cheby1', edges, 2, 0.01, 20.0)[0][1] is False


def test_filter_responses_are_scipys_bit_for_bit():
    """planner.sosfreqz_bands restates scipy.signal.sosfreqz (freqz_sos -> freqz -> polyval) with the same NumPy
    operations on arrays of the same shape: w and h must carry SciPy's bits for every band, filter type, order and kind
    of frequency grid; inputs it does not cover are declined (None) and go to SciPy."""
    from scipy import signal
    fr = np.logspace(-2, np.log10(20.0), 1000)
    n = 0
    for ftype in ('butter', 'cheby1'):
        for order in (1, 2, 4):
            for fs in (20.0, 40.0, 37.3):
                edges = np.logspace(np.log10(0.05), np.log10(fs * 0.49), 13)
                sos_list = [signal.iirfilter(order, [edges[i], edges[i + 1]], rp=0.01, btype='band', ftype=ftype, output='sos', fs=fs)
                            for i in range(12)]
                for worN in (fr, np.linspace(0.0, fs / 2, 333), np.arange(1, 9)):
                    w, rows = planner.sosfreqz_bands(sos_list, worN, fs)
                    for i, sos in enumerate(sos_list):
                        ws, hs = signal.sosfreqz(sos, worN, fs=fs)
                        assert np.array_equal(ws.view(np.uint64), w.view(np.uint64))
                        assert np.array_equal(hs.view(np.uint64), rows[i].view(np.uint64)), (ftype, order, fs, i)
                        n += 1
    assert n > 600
    assert planner.sosfreqz_bands([np.zeros((2, 6))], 512, 40.0) is None               # integer worN: SciPy's FFT branch
    assert planner.sosfreqz_bands([np.zeros((0, 6))], fr, 40.0) is None
    assert planner.sosfreqz_bands([np.zeros((2, 6), dtype=np.float32)], fr, 40.0) is None


def test_plan_constants_are_kept_read_only_and_equal_a_fresh_computation():
    """planner keeps the constants that depend on (pair count, alpha), the trace length or the geometry between calls
    (LCG subsets, factor tables, taper ramps, co-array).  What comes back must be what a fresh computation gives, must
    not be writable (it is shared between calls), and a different argument must not hit a stale entry."""
    from narrow_band_least_squares_amd import planner, synthetic
    rij = synthetic.array_geometry(8, 1.0, seed=3)
    a = planner.co_array(rij)
    b = planner.co_array(rij.copy())
    assert all(x is y for x, y in zip(a, b)) and not any(x.flags.writeable for x in a)
    fresh = planner._co_array(np.ascontiguousarray(rij, dtype=np.float64))
    for x, y in zip(a, fresh):
        np.testing.assert_array_equal(x, y)
    other = planner.co_array(rij * 1.5)
    assert not np.array_equal(other[0], a[0])
    for k in range(12):                                   # more geometries than the cache keeps: the first one is recomputed, equal
        planner.co_array(rij + k)
    np.testing.assert_array_equal(planner.co_array(rij)[2], fresh[2])
    tl, tr = planner.taper_ramps(4000)
    tl2, tr2 = planner.taper_ramps(4000)
    assert tl is tl2 and not tl.flags.writeable and len(tl) == int(planner.TAPER_FRACTION * 4000)
    assert len(planner.taper_ramps(4400)[0]) == int(planner.TAPER_FRACTION * 4400)
    p1 = planner.lts_plan(a[0], 0.5)
    p2 = planner.lts_plan(a[0], 0.5)
    np.testing.assert_array_equal(p1['rew_table'], p2['rew_table'])
    assert p1['rew_table'] is not p2['rew_table'] and p1['raw_factor'] == p2['raw_factor']
    p1['rew_table'][0] = 123.0                            # the caller's copy is its own
    assert planner.lts_plan(a[0], 0.5)['rew_table'][0] == 1.0
    assert planner.lts_plan(a[0], 0.75)['h'] != p1['h']
