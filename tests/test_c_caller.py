"""The C ABI from outside the package: (1) a plain C program compiled against include/nbls.h and linked with
libnbls_hip.so (tests/c_caller/nbls_caller.c) — compile + link on the CPU box, run on the GPU box; (2) the
ctypes stub printed in INTEGRATION.md, executed verbatim.  Both go through nbls_run() and are compared with
the oracle (lags / weights exact, vel / baz 1e-9) and with the package's own path (bit identical)."""
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CDIR = os.path.join(ROOT, 'tests', 'c_caller')
BIN = os.path.join(CDIR, 'nbls_caller')
LIBDIR = os.path.join(ROOT, 'narrow_band_least_squares_amd', 'csrc')


def build_caller():
    cmd = ['gcc', '-O1', '-Wall', '-Wextra', '-Werror', '-std=c11', '-pthread', '-I', os.path.join(ROOT, 'include'),
           os.path.join(CDIR, 'nbls_caller.c'), '-o', BIN, '-L', LIBDIR, '-lnbls_hip',
           '-Wl,-rpath,$ORIGIN/../../narrow_band_least_squares_amd/csrc', '-Wl,-rpath-link,/opt/rocm/lib']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return BIN


def test_c_caller_compiles_and_links_against_the_header():
    """-Wall -Wextra -Werror, C11: the header is plain C and declares everything the caller uses."""
    build_caller()
    out = subprocess.run(['nm', '-u', BIN], capture_output=True, text=True).stdout
    used = set(re.findall(r'\b(nbls_[a-z0-9_]+)', out))
    assert {'nbls_create', 'nbls_set_trace', 'nbls_set_geometry', 'nbls_run', 'nbls_plan', 'nbls_destroy',
            'nbls_last_error', 'nbls_version'} <= used


def _problem(oracle, alpha):
    from narrow_band_least_squares_amd import planner, synthetic
    c = synthetic.build_config('cfg2', 0.12)
    data = np.ascontiguousarray(c['data'])
    fs = c['fs']
    xij, pair_idx, xpinv = planner.co_array(c['rij'])
    edges = [(0.3, 1.0), (1.0, 3.0)]
    sos = planner.pad_sections([planner.design_bandpass('butter', lo, hi, 2, 0.01, fs)[0] for lo, hi in edges])
    tl, tr = planner.taper_ramps(data.shape[1])
    W, inc, nwin = planner.window_plan(data.shape[1], fs, 30.0, 0.5)
    lts = planner.lts_plan(xij, alpha) if alpha < 1.0 else None
    return dict(c=c, data=data, fs=fs, xij=xij, pair_idx=pair_idx, xpinv=xpinv, edges=edges, sos=sos, tl=tl, tr=tr,
                W=np.array([W, W], dtype=np.int32), inc=np.array([inc, inc], dtype=np.int32), nwin=nwin, lts=lts,
                alpha=alpha, vector_len=nwin + 2)


def _expected(oracle, p):
    out = []
    for lo, hi in p['edges']:
        st = oracle.make_stream(p['data'], p['fs'])
        stf, _, _ = oracle.filter_data(st, 'butter', lo, hi, 2, 0.01)
        res, internals = oracle.ltsva(stf, None, None, 30.0, 0.5, p['alpha'], rij=p['c']['rij'], return_internals=True)
        out.append((res, internals))
    return out


def _check(p, exp, vel, baz, mdccm, nwin, lag=None, weights=None):
    n = p['nwin']
    assert list(nwin) == [n, n]
    for b, (res, internals) in enumerate(exp):
        np.testing.assert_allclose(vel[b, :n], res[0], rtol=1e-9)
        np.testing.assert_allclose(baz[b, :n], res[1], rtol=1e-9)
        np.testing.assert_allclose(mdccm[b, :n], res[3], rtol=1e-12)
        assert not vel[b, n:].any()
        if lag is not None:
            np.testing.assert_array_equal(lag[b, :n], np.rint(internals['tau'].T * p['fs']).astype(int))
        if weights is not None and p['alpha'] < 1.0:
            np.testing.assert_array_equal(weights[b, :n], internals['weights'].T)


@pytest.mark.gpu
@pytest.mark.parametrize('alpha', [1.0, 0.5])
def test_c_program_runs_nbls_run(oracle, tmp_path, alpha):
    p = _problem(oracle, alpha)
    if not os.path.exists(BIN):
        build_caller()
    B, S, VL, P = 2, p['sos'].shape[1], p['vector_len'], len(p['xij'])
    nch, npts = p['data'].shape
    blob = struct.pack('<8i', nch, P, B, S, 1, len(p['tl']), VL, int(alpha < 1.0)) + struct.pack('<q', npts) + struct.pack('<d', p['fs'])
    blob += p['data'].tobytes() + p['xij'].tobytes() + p['pair_idx'].astype(np.int32).tobytes() + p['xpinv'].tobytes()
    blob += p['sos'].tobytes() + p['tl'].tobytes() + p['tr'].tobytes() + p['W'].tobytes() + p['inc'].tobytes()
    if alpha < 1.0:
        l = p['lts']
        st = np.ascontiguousarray(l['starts'], dtype=np.int32)
        blob += struct.pack('<d', l['alpha']) + struct.pack('<5i', l['h'], st.shape[0], l['csteps'], l['csteps2'], l['ncand'])
        blob += st.tobytes() + np.asarray(l['xij_mad'], dtype=np.float64).tobytes() + struct.pack('<d', l['raw_factor'])
        blob += np.asarray(l['rew_table'], dtype=np.float64).tobytes() + struct.pack('<2d', l['quantile'], l['zero_scale'])
    prob, res = str(tmp_path / 'problem.bin'), str(tmp_path / 'result.bin')
    open(prob, 'wb').write(blob)
    r = subprocess.run([BIN, prob, res], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'C_CALLER_OK' in r.stdout, r.stdout + r.stderr
    raw = open(res, 'rb').read()
    cells = B * VL
    grids = np.frombuffer(raw, dtype=np.float64, count=4 * cells).reshape(4, B, VL)
    off = 32 * cells
    nwin = np.frombuffer(raw, dtype=np.int32, count=B, offset=off); off += 4 * B
    lag = np.frombuffer(raw, dtype=np.int32, count=cells * P, offset=off).reshape(B, VL, P); off += 4 * cells * P
    wts = np.frombuffer(raw, dtype=np.uint8, count=cells * P, offset=off).reshape(B, VL, P)
    _check(p, _expected(oracle, p), grids[0], grids[1], grids[2], nwin, lag, wts)
    # bit identical to the package's own ctypes path on the same inputs
    from narrow_band_least_squares_amd import engine
    mine = engine.process(p['data'], p['fs'], 0.0, p['c']['rij'], p['edges'], [30.0, 30.0], 0.5, alpha, 'butter', 2, 0.01,
                          vector_len=VL, want_lag=True)
    np.testing.assert_array_equal(mine.vel, grids[0])
    np.testing.assert_array_equal(mine.baz, grids[1])
    np.testing.assert_array_equal(mine.lag, lag)


@pytest.mark.gpu
def test_integration_md_stub_verbatim(oracle):
    """The ctypes stub a maintainer would paste into the reference, executed as printed."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'<!-- stub:begin.*?-->\s*```python\n(.*?)```\s*<!-- stub:end -->', text, re.S)
    assert m, 'stub markers not found in INTEGRATION.md'
    p = _problem(oracle, 1.0)
    ns = dict(LIBNBLS_PATH=os.path.join(LIBDIR, 'libnbls_hip.so'), st=oracle.make_stream(p['data'], p['fs']), fs=p['fs'],
              xij=p['xij'], pair_idx=np.ascontiguousarray(p['pair_idx'], dtype=np.int32), xpinv=p['xpinv'], sos=p['sos'],
              zero_phase=1, tl=p['tl'], tr_=p['tr'], W=p['W'], inc=p['inc'], vector_len=p['vector_len'], lts_params=None)
    exec(compile(m.group(1), 'INTEGRATION.md', 'exec'), ns)
    _check(p, _expected(oracle, p), ns['vel'], ns['baz'], ns['mdccm'], ns['nwin'])
    assert ns['weights'][0, :p['nwin']].all()          # OLS: every pair kept
