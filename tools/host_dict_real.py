"""Developer: host cost of the stdict dictionary on the REAL weight masks of one cfg-3 call."""
import contextlib
import io
import sys
import time

sys.path.insert(0, '/root/repo')
import numpy as np
from narrow_band_least_squares_amd import engine, synthetic, planner

c = synthetic.build_config(sys.argv[1] if len(sys.argv) > 1 else 'cfg3', 1.0)
rows, fs, t0 = engine.stream_rows(c['st'])
edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(c['NBANDS'])]
with contextlib.redirect_stdout(io.StringIO()):
    res = engine.process(rows, fs, t0, c['rij'], edges, c['WINLEN_list'], c['overlap'], c['alpha'], c['ftype'], c['order'], c['ripple'])
mask, nwin = res.mask, res.nwin
B = len(nwin)
codes = mask.reshape(-1, mask.shape[2])
valid = (np.arange(mask.shape[1])[None, :] < np.asarray(nwin)[:, None]).ravel()
u = np.unique(codes[valid], axis=0)
print('units', int(valid.sum()), 'distinct weight patterns', len(u))
pref = ['%02d_' % (b + 1) for b in range(B)]
for rep in range(5):
    t0_ = time.perf_counter()
    keys = engine.time_keys(res.t, nwin, pref)
    t1 = time.perf_counter()
    d = engine.new_stdict(len(keys))
    t2 = time.perf_counter()
    engine.stdict_from_mask(mask, nwin, res.pair_idx, res.nchans, keys, d, 0)
    t3 = time.perf_counter()
    print('time_keys %.2f ms  new_dict %.2f ms  stdict %.2f ms (%d entries, %.0f ns/entry)'
          % ((t1 - t0_) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(d), (t3 - t2) * 1e9 / len(d)))
    del d, keys
