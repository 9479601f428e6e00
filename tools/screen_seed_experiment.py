"""Developer experiment (r04): how much of the screening kernel's time would ANY scheme save that knows
the correlation maximum before the first round of lag groups (a scout pass over lags 0..127, a K-split first round)?
The developer build's option screen_seed starts every (unit, sliding channel, partner)'s running maximum from the
screening maximum the previous pass over the same unit batch left in the candidate records, and lets the first round
prune.  ONE unit batch (scale 0.14 of cfg-3: 9 600 units), plane wave and incoherent noise.

    NBLS_LIB=narrow_band_least_squares_amd/csrc/libnbls_hip_dev.so python tools/screen_seed_experiment.py"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit('/', 2)[0])
from narrow_band_least_squares_amd import engine, synthetic  # noqa: E402

c = synthetic.build_config('cfg3', scale=0.14)
h = engine.get_handle()
assert h.lib.nbls_developer_build(), 'needs the developer build (NBLS_LIB=.../libnbls_hip_dev.so)'
edges = [(c['freqlist'][i], c['freqlist'][i + 1]) for i in range(c['NBANDS'])]
for label, rows in (('plane wave, SNR 6 dB', engine.stream_rows(c['st'])[0]),
                    ('incoherent noise', [np.random.default_rng(7 + i).standard_normal(c['npts']) for i in range(c['N'])])):
    prep = engine.prepare(len(rows), len(rows[0]), c['fs'], c['rij'], edges, c['WINLEN_list'], c['overlap'], c['alpha'], c['ftype'], c['order'], c['ripple'])
    h.set_profiling(True)
    out = {}
    for seed in (0, 1):
        h.set_option('screen_seed', seed)
        engine.launch(h, rows, prep)
        h.sync()
        assert h.timings()['xcorr_launches'] == 1, 'one unit batch expected'
        ts = []
        for _ in range(5):
            h.execute(); h.sync()
            ts.append(h.timings()['screen_ms'])
        out[seed] = (np.median(ts), h.fetch(want_lag=True)['lag'].copy())
    h.set_option('screen_seed', 0)
    assert np.array_equal(out[0][1], out[1][1]), 'lags differ'
    print('%-22s units %d: screen_kernel %.3f ms; maxima known from the start %.3f ms (%.1f %% less); lags identical'
          % (label, int(prep.nwin.sum()), out[0][0], out[1][0], 100 * (1 - out[1][0] / out[0][0])))
