"""Developer: per-call wall times of the whole cfg-3 call (jitter, warm-up drift); NBLS_PIPELINE_GROUPS / NBLS_STREAM_RESULTS
from the environment."""
import contextlib, io, os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic, planner
from narrow_band_least_squares_amd import engine
pos = [a for a in sys.argv[1:] if '=' not in a]
for a in sys.argv[1:]:
    if '=' in a:
        engine.get_handle().set_option(a.split('=')[0], int(a.split('=')[1]))
c = synthetic.build_config(pos[0] if pos else 'cfg3', 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000); w = np.zeros(1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr,
        c['ftype'], c['order'], c['ripple'])
ts = []
for rep in range(30):
    planner.design_cache_clear()
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        out = narrow_band_least_squares(*args, rij=c['rij'])
    ts.append((time.perf_counter() - t) * 1e3)
    del out
print(' '.join('%.1f' % x for x in ts))
print('stream=%s ' % os.environ.get('NBLS_STREAM_RESULTS', '1') + 'groups=%s split=%s: median %.2f  min %.2f  max %.2f ms' % (os.environ.get('NBLS_PIPELINE_GROUPS'), engine.PIPELINE_SPLIT,
                                                                   np.median(ts[5:]), min(ts[5:]), max(ts[5:])))
