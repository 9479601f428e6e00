import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from narrow_band_least_squares_amd import engine, synthetic
c = synthetic.build_config('cfg3', 0.2)
data, fs, t0 = engine.stream_to_array(c['st'])
h = engine.get_handle()
for b in (0, 6, 12, 24, 36, 47):
    e = [(c['freqlist'][b], c['freqlist'][b+1])]
    engine.process(data, fs, t0, c['rij'], e, [30.0], 0.5, 0.5, 'butter', 2, 0.01, xcorr_impl=3)
    print(b, e, h.screen_stats(), flush=True)
