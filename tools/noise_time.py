import sys, time
sys.path.insert(0, '.')
import numpy as np
from narrow_band_least_squares_amd import engine, synthetic, planner
c = synthetic.build_config('cfg3', 0.25)
rng = np.random.default_rng(3)
for label, data in (('coherent', c['data']), ('pure noise', rng.standard_normal(c['data'].shape)),
                    ('one dead + noise', np.vstack([np.zeros((1, c['data'].shape[1])), rng.standard_normal((7, c['data'].shape[1]))]))):
    edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(c['NBANDS'])]
    h = engine.get_handle(); h.set_profiling(True)
    for rep in range(2):
        res = engine.process(data, c['fs'], 0.0, c['rij'], edges, c['WINLEN_list'], c['overlap'], c['alpha'], c['ftype'], c['order'], c['ripple'])
    tm = h.timings()
    print(label, 'units', int(res.nwin.sum()), {k: round(v, 2) for k, v in tm.items() if k.endswith('_ms')}, h.screen_stats())
