import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from narrow_band_least_squares_amd import engine, synthetic
name = sys.argv[1]; scale=float(sys.argv[2]); nb=int(sys.argv[3])
c = synthetic.build_config(name, scale)
data, fs, t0 = engine.stream_to_array(c['st'])
edges = [(c['freqlist'][i], c['freqlist'][i+1]) for i in range(0, c['NBANDS'], max(1, c['NBANDS']//nb))][:nb]
wl = [c['WINLEN_list'][0]]*len(edges)
h = engine.get_handle(); h.set_profiling(True)
res = {}
for impl in (3, 2, 1):
    try:
        t=time.time()
        r = engine.process(data, fs, t0, c['rij'], edges, wl, 0.5, c['alpha'], 'butter', 2, 0.01, xcorr_impl=impl, want_lag=True, want_cmax=True)
        print(impl, 'units', int(r.nwin.sum()), 'wall %.3f'%(time.time()-t), h.timings(), flush=True)
        if impl == 3: print(h.screen_stats(), flush=True)
        res[impl] = r
    except Exception as e:
        print(impl, 'ERR', e, flush=True)
ks = sorted(res)
for k in ks[1:]:
    print('lags equal', ks[0], k, np.array_equal(res[ks[0]].lag, res[k].lag), 'cmax maxdiff', np.nanmax(np.abs(res[ks[0]].cmax - res[k].cmax)))
