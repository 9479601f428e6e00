"""Developer: one-off soak of ltsva() with 9..24 elements (the large-array FAST-LTS kernel, partner-group screening)
against the oracle: lags, weights, z exactly / to 1e-9.    python tools/soak_large.py FIRST LAST [NLO NHI]
(NLO NHI = 24 33: the u16-counter / no-merging form of the large-array LTS kernel, more than 255 pairs)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import contextlib, io
import numpy as np
import nbls_oracle as oracle
import test_gpu_parity as T
from narrow_band_least_squares_amd import synthetic
first, last = int(sys.argv[1]), int(sys.argv[2])
nlo, nhi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (9, 25)      # element counts [nlo, nhi)
bad = 0
for seed in range(first, last):
    rng = np.random.default_rng(5000 + seed)
    nchans = int(rng.integers(nlo, nhi))
    fs = float(rng.choice([20.0, 40.0]))
    winlen = float(rng.choice([15.0, 20.0, 30.0]))
    npts = int((rng.uniform(5.0, 9.0) * winlen) * fs)
    alpha = float(rng.choice([0.5, 0.5, 0.6, 0.75, 0.9]))
    rij = synthetic.array_geometry(nchans, float(rng.uniform(0.5, 2.0)), seed=int(rng.integers(1 << 30)))
    nbad = int(rng.integers(0, 3))
    data = synthetic.plane_wave(rij, npts, fs, 0.3, 0.4 * fs, baz_deg=float(rng.uniform(0, 360)), vel_kms=float(rng.uniform(0.3, 3.0)),
                                snr_db=float(rng.uniform(-6, 12)), timing_error_s=0.3 if nbad else 0.0,
                                bad_element=nchans - 1 if nbad else None, seed=int(rng.integers(1 << 30)))
    if nbad == 2:
        data[0] = np.roll(data[0], 7)
    c = dict(fs=fs, rij=rij - rij.mean(axis=1, keepdims=True))
    st = oracle.make_stream(data, fs, starttime=17884.0729166667)
    stf, _, _ = oracle.filter_data(st, 'butter', 0.5, 0.35 * fs, 2, 0.01)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            T._compare_ltsva(oracle, c, stf, winlen, alpha)
    except Exception as e:      # noqa: BLE001
        bad += 1
        print('seed', seed, 'nchans', nchans, 'alpha', alpha, 'FAILED:', type(e).__name__, str(e)[:300], flush=True)
    if (seed - first) % 10 == 9:
        print('... up to seed', seed, 'failures so far', bad, flush=True)
print('soak done: seeds %d..%d, %d failures' % (first, last - 1, bad))
