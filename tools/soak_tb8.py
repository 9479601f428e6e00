"""Developer: one-off soak of the eight-tile instance of the screening kernel (10..24 elements, windows of 1800..4500
samples, with and without partner groups, coherent and incoherent data): lags of the int8 screening path against the
plain VALU correlator, exactly.    python tools/soak_tb8.py FIRST LAST"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from narrow_band_least_squares_amd import engine, synthetic
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = skipped = 0
for seed in range(first, last):
    rng = np.random.default_rng(9000 + seed)
    nchans = int(rng.integers(10, 25))
    fs = 40.0
    W = int(rng.integers(1800, 4500 if nchans <= 17 else 2900))      # (18+ elements: partner groups of 16, 2 x 16 images)
    winlen = W / fs
    npts = int(rng.uniform(2.2, 3.5) * W)
    rij = synthetic.array_geometry(nchans, float(rng.uniform(0.5, 2.0)), seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.4:
        data = rng.standard_normal((nchans, npts))
    else:
        data = synthetic.plane_wave(rij, npts, fs, 0.2, 8.0, baz_deg=float(rng.uniform(0, 360)), vel_kms=float(rng.uniform(0.3, 3.0)),
                                    snr_db=float(rng.uniform(-6, 12)), seed=int(rng.integers(1 << 30)))
    f0 = float(rng.choice([0.05, 0.2, 1.0, 3.0]))
    edges = [(f0, f0 * float(rng.choice([1.3, 2.0, 4.0])))]
    kw = dict(want_lag=True, want_cmax=True)
    try:
        ref = engine.process(data, fs, 0.0, rij, edges, [winlen], 0.5, 1.0, 'butter', 2, 0.01, xcorr_impl=1, **kw)
        got = engine.process(data, fs, 0.0, rij, edges, [winlen], 0.5, 1.0, 'butter', 2, 0.01, xcorr_impl=3, **kw)
        tm = engine.get_handle().timings()
        np.testing.assert_array_equal(got.lag, ref.lag)
        np.testing.assert_allclose(got.cmax, ref.cmax, rtol=1e-12, atol=1e-15)
        print('seed %d ok: %d elements, W %d, %d windows, band %.2f-%.2f Hz, impl %d' % (seed, nchans, W, int(got.nwin[0]), edges[0][0], edges[0][1], tm['xcorr_impl']), flush=True)
    except ValueError as e:
        if "fit a CU's LDS" not in str(e):
            raise
        skipped += 1
        print('seed %d skipped: %d elements, W %d: images do not fit LDS' % (seed, nchans, W), flush=True)
    except Exception as e:      # noqa: BLE001
        bad += 1
        print('seed', seed, 'FAILED:', nchans, W, type(e).__name__, str(e)[:300], flush=True)
print('soak done: seeds %d..%d, %d failures, %d skipped' % (first, last - 1, bad, skipped))
