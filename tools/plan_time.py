"""Developer: host cost of nbls_plan (developer build: the plan_timing option prints the phases to stderr).
    NBLS_LIB=.../libnbls_hip_dev.so python tools/plan_time.py [cfg] [bands]"""
import sys
import time

sys.path.insert(0, '/root/repo')
import numpy as np
from narrow_band_least_squares_amd import engine, synthetic

c = synthetic.build_config(sys.argv[1] if len(sys.argv) > 1 else 'cfg3', 1.0)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else c['NBANDS']
rows, fs, t0 = engine.stream_rows(c['st'])
edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(nb)]
prep = engine.prepare(len(rows), len(rows[0]), fs, c['rij'], edges, c['WINLEN_list'][:nb], 0.5, c['alpha'], c['ftype'], 2, 0.01)
h = engine.get_handle()
if h.lib.nbls_developer_build():
    h.set_option('plan_timing', 1)
for rep in range(4):
    t = time.perf_counter()
    engine.launch(h, rows, prep)
    t1 = time.perf_counter()
    h.sync()
    print('launch (upload + geometry + plan + execute) %.2f ms, pass %.2f ms' % ((t1 - t) * 1e3, (time.perf_counter() - t1) * 1e3))
