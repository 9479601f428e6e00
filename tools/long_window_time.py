"""Developer: correlation-stage time of long windows — screening correlator with partner groups against the general
correlators it used to fall back to.    python tools/long_window_time.py"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit('/', 2)[0])
from narrow_band_least_squares_amd import engine, synthetic  # noqa: E402

fs = 100.0
h = engine.get_handle()
h.set_profiling(True)
for nchans, W in ((8, 6000), (16, 3200), (16, 4500), (8, 7500), (8, 9000), (8, 12000), (16, 9000), (5, 15000)):
    rij = synthetic.array_geometry(nchans, 1.0, seed=nchans)
    data = synthetic.plane_wave(rij, 60 * W // 2 + W, fs, 0.5, 20.0, seed=3)
    for impl in (0, 2, 1):            # 0 = what a call gets: screening where its images fit (<= ~13 000 samples), else a general correlator
        try:
            r = engine.process(data, fs, 0.0, rij, [(0.5, 20.0)], [W / fs + 1e-9], 0.5, 1.0, 'butter', 2, 0.01, xcorr_impl=impl)
            h.execute(); h.sync()
            tm = h.timings()
            print('%2d elements, W = %d, %d windows: xcorr_impl %d -> correlation stage %.2f ms (used %d)'
                  % (nchans, W, int(r.nwin[0]), impl, tm['xcorr_ms'], tm['xcorr_impl']), flush=True)
        except Exception as e:      # noqa: BLE001
            print('%2d elements, W = %d: xcorr_impl %d not available (%s)' % (nchans, W, impl, str(e)[:60]), flush=True)
