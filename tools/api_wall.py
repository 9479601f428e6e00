import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo' if len(sys.argv) < 2 else sys.argv[1])
import numpy as np
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic
c = synthetic.build_config('cfg3', 1.0)
w = np.zeros(1000); h = np.zeros(1000); fr = np.linspace(0.01, 20, 1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, h, c['freqlist'],
        c['band_type'], fr, c['ftype'], c['order'], c['ripple'])
for rep in range(3):
    t = time.time(); out = narrow_band_least_squares(*args, rij=c['rij']); print('wall %.3f s' % (time.time() - t), len(out[4]))
pr = cProfile.Profile(); pr.enable(); out = narrow_band_least_squares(*args, rij=c['rij']); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
