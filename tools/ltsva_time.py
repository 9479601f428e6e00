"""Developer: wall time of the broadband route example.py:108-109 takes — filter_data(st, ...) then ltsva(stf, ...) — on a
named configuration's trace.    python tools/ltsva_time.py [cfg3] [winlen_s] [alpha]"""
import contextlib, io, sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import numpy as np
from narrow_band_least_squares_amd import filter_data, ltsva, synthetic

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
winlen = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
alpha = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
c = synthetic.build_config(name, 1.0)
st = c['st']
fmin, fmax = c['freqlist'][0], c['freqlist'][-1]
tf, tl = [], []
for rep in range(12):
    t0 = time.perf_counter()
    stf, fs, sos = filter_data(st, c['ftype'], fmin, fmax, c['order'], c['ripple'])
    t1 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        out = ltsva(stf, None, None, winlen, 0.5, alpha, False, rij=c['rij'])
    t2 = time.perf_counter()
    tf.append((t1 - t0) * 1e3)
    tl.append((t2 - t1) * 1e3)
print('%s: %d windows of %.0f s, alpha %.2f' % (name, len(out[0]), winlen, alpha))
print('filter_data ms:', ' '.join('%.2f' % x for x in tf))
print('ltsva ms      :', ' '.join('%.2f' % x for x in tl))
print('median filter_data %.2f ms, ltsva %.2f ms; vel_uncert median %.4g km/s, baz_uncert median %.4g deg' %
      (np.median(tf[3:]), np.median(tl[3:]), np.nanmedian(out[6]), np.nanmedian(out[7])))
