import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
import nbls_oracle as o
from narrow_band_least_squares_amd import synthetic, helpers
c = synthetic.build_config('cfg2', 0.3)
for ftype in ('butter', 'cheby1'):
    for order in (1, 2, 3, 4, 5, 6, 8):
        try:
            stf, fs, sos = helpers.filter_data(c['st'], ftype, 0.8, 3.0, order, 0.01)
            st_o = o.make_stream(c['data'], c['fs'], starttime=c['st'][0].stats.starttime)
            stf_o, fs_o, sos_o = o.filter_data(st_o, ftype, 0.8, 3.0, order, 0.01)
            scale = max(np.abs(tr.data).max() for tr in stf_o)
            err = max(np.max(np.abs(a.data - b.data)) for a, b in zip(stf, stf_o)) / scale
            print(ftype, order, 'sections', sos.shape[0], 'max err / scale %.2e' % err)
        except Exception as e:
            print(ftype, order, 'ERR', type(e).__name__, e)
