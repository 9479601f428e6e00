import sys, time
sys.path.insert(0, '.')
import numpy as np
from narrow_band_least_squares_amd import engine, synthetic, planner
c = synthetic.build_config('cfg3', 1.0)
data, fs, t0 = engine.stream_to_array(c['st'])
edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(c['NBANDS'])]
import cProfile, pstats
engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], 0.5, 0.5, 'butter', 2, 0.01)
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], 0.5, 0.5, 'butter', 2, 0.01)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
