import contextlib, gc, io, os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic, planner
c = synthetic.build_config('cfg3', 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000); w = np.zeros(1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr, c['ftype'], c['order'], c['ripple'])
def call():
    planner.design_cache_clear()
    with contextlib.redirect_stdout(io.StringIO()):
        return narrow_band_least_squares(*args, rij=c['rij'])
for mode in ('plain', 'freeze', 'plain', 'freeze'):
    for _ in range(3): call()
    if mode == 'freeze':
        gc.collect(); gc.freeze()
    else:
        gc.unfreeze()
    held, ts = [], []
    c0 = gc.get_stats()[2]['collections']
    for rep in range(60):
        t = time.perf_counter(); held.append(call()); ts.append((time.perf_counter() - t) * 1e3)
    print('%-7s held results, 60 calls: mean %.2f median %.2f max %.2f ms; gen-2 collections %d' % (mode, np.mean(ts), np.median(ts), max(ts), gc.get_stats()[2]['collections'] - c0))
    del held
