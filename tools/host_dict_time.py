"""Developer: host cost of the stdict keys / dictionary of one cfg-3 call (no GPU needed)."""
import sys
import time

sys.path.insert(0, '/root/repo')
import numpy as np
from narrow_band_least_squares_amd import engine

B, VL, P = 48, 1438, 28
rng = np.random.default_rng(0)
nwin = np.full(B, VL, dtype=np.int64)
t = 19000.0 + np.arange(B * VL).reshape(B, VL) * (15.0 / 86400)
pref = ['%02d_' % (b + 1) for b in range(B)]
mask = np.full((B, VL, 4), 0xff, dtype=np.uint8)
mask[rng.random((B, VL)) < 0.7, 0] = 0x0f
mask[rng.random((B, VL)) < 0.1, 2] = 0xfe
pair_idx = np.array([(i, j) for i in range(8) for j in range(i + 1, 8)], dtype=np.int32)
for rep in range(6):
    t0 = time.perf_counter()
    keys = engine.time_keys(t, nwin, pref)
    t1 = time.perf_counter()
    d = engine.new_stdict(B * VL)
    t2 = time.perf_counter()
    for g in range(4):
        b0, b1 = 12 * g, 12 * g + 12
        engine.stdict_from_mask(mask[b0:b1], nwin[b0:b1], pair_idx, 8, keys, d, b0 * VL)
    t3 = time.perf_counter()
    print('time_keys %.2f ms (%d keys, %.0f ns/key)  new_dict %.2f ms  stdict %.2f ms (%d entries, %.0f ns/entry)'
          % ((t1 - t0) * 1e3, len(keys), (t1 - t0) * 1e9 / len(keys), (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(d), (t3 - t2) * 1e9 / len(d)))
    del d, keys
    for nt in (1, 2, 4, 8):
        import os
        engine.KEY_THREADS = nt
        t0 = time.perf_counter()
        kt = engine.time_key_text(t, nwin, pref)
        t1 = time.perf_counter()
        d = engine.new_stdict(B * VL)
        t2 = time.perf_counter()
        for g in range(4):
            b0, b1 = 12 * g, 12 * g + 12
            engine.stdict_from_mask(mask[b0:b1], nwin[b0:b1], pair_idx, 8, kt, d, b0 * VL)
        t3 = time.perf_counter()
        print('   text keys, %d threads: key text %.2f ms  stdict %.2f ms (%d entries)' % (nt, (t1 - t0) * 1e3, (t3 - t2) * 1e3, len(d)))
        del d, kt
