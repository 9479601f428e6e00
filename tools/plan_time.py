"""Developer: host cost of nbls_plan at cfg-3 (NBLS_PLAN_TIMING=1 prints the phases to stderr)."""
import os, sys, time
os.environ['NBLS_PLAN_TIMING'] = '1'
sys.path.insert(0, '/root/repo')
import numpy as np
from narrow_band_least_squares_amd import engine, synthetic
c = synthetic.build_config(sys.argv[1] if len(sys.argv) > 1 else 'cfg3', 1.0)
rows, fs, t0 = engine.stream_rows(c['st'])
edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(c['NBANDS'])]
prep = engine.prepare(len(rows), len(rows[0]), fs, c['rij'], edges, c['WINLEN_list'], 0.5, c['alpha'], c['ftype'], 2, 0.01)
h = engine.get_handle()
for rep in range(4):
    t = time.perf_counter(); engine.launch(h, rows, prep); t1 = time.perf_counter(); h.sync()
    print('launch %.2f ms, pass %.2f ms' % ((t1 - t) * 1e3, (time.perf_counter() - t1) * 1e3))
