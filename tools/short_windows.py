import sys
sys.path.insert(0, '.')
import numpy as np
from narrow_band_least_squares_amd import engine, synthetic
for nch, fs, wl in ((5, 40.0, 1.0), (8, 20.0, 2.5), (3, 100.0, 0.5), (16, 20.0, 3.2)):
    rij = synthetic.array_geometry(nch, 1.0, seed=nch)
    data = synthetic.plane_wave(rij, 2000, fs, 0.5, 0.4 * fs, seed=2)
    h = engine.get_handle(); h.set_profiling(True)
    kw = dict(want_lag=True, want_cmax=True)
    a = engine.process(data, fs, 0.0, rij, [(1.0, 4.0)], [wl], 0.5, 1.0, 'butter', 2, 0.01, **kw)
    impl = h.timings()['xcorr_impl']
    b = engine.process(data, fs, 0.0, rij, [(1.0, 4.0)], [wl], 0.5, 1.0, 'butter', 2, 0.01, xcorr_impl=1, **kw)
    print(nch, 'W', int(wl * fs), 'impl', impl, 'lags equal', np.array_equal(a.lag, b.lag), 'nwin', int(a.nwin[0]))
