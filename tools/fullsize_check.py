"""One-off validation at production size: GPU cfg-3 (all 48 bands) against the oracle on a few bands
(lags, weights exact; vel/baz/mdccm to 1e-9).  Developer tool (the oracle needs ~1 min per band)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
import nbls_oracle as o
from narrow_band_least_squares_amd import engine, synthetic

c = synthetic.build_config('cfg3', 1.0)
edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(c['NBANDS'])]
data, fs, t0 = engine.stream_to_array(c['st'])
res = engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], c['overlap'], c['alpha'], c['ftype'], c['order'],
                     c['ripple'], want_lag=True, want_z=True)
for b in [int(x) for x in sys.argv[1:]] or [0, 23, 47]:
    t = time.time()
    st = o.make_stream(c['data'], fs, starttime=c['st'][0].stats.starttime)
    stf, _, _ = o.filter_data(st, 'butter', edges[b][0], edges[b][1], 2, 0.01)
    out, it = o.ltsva(stf, None, None, 30.0, 0.5, c['alpha'], rij=c['rij'], return_internals=True)
    n = int(res.nwin[b])
    lag_o = np.rint(it['tau'].T * fs).astype(int)
    same_lag = np.array_equal(res.lag[b, :n], lag_o)
    same_w = np.array_equal(res.weights[b, :n], it['weights'].T)
    dv = np.nanmax(np.abs(res.vel[b, :n] / out[0] - 1))
    db = np.nanmax(np.abs(res.baz[b, :n] - out[1]))
    dm = np.nanmax(np.abs(res.mdccm[b, :n] / out[3] - 1))
    print('band %d: windows %d lags equal %s weights equal %s max rel dvel %.2e max dbaz %.2e max rel dmdccm %.2e (%.0f s)'
          % (b, n, same_lag, same_w, dv, db, dm, time.time() - t), flush=True)
