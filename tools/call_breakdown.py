"""Developer: where the wall time of one whole narrow_band_least_squares() call goes (cfg-3 by default).
Wraps the host-side phases with perf_counter; prints mean ms per phase over a few calls."""
import collections
import functools
import sys
import time

sys.path.insert(0, '/root/repo' if len(sys.argv) < 3 else sys.argv[2])
import numpy as np
from scipy import signal
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic, engine, planner, _hip

acc = collections.defaultdict(float)


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    @functools.wraps(f)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
    setattr(obj, name, g)


for n in ('stream_rows', 'prepare', 'launch', 'all_window_times', 'time_keys', 'stdict_from_mask'):
    wrap(engine, n)
for n in ('co_array', 'design_bandpass', 'lts_plan', 'taper_ramps', 'pad_sections'):
    wrap(planner, n, 'prepare.' + n)
for n in ('set_trace_rows', 'set_geometry', 'plan', 'execute', 'fetch_packed'):
    wrap(_hip.Handle, n, 'handle.' + n)
wrap(signal, 'sosfreqz')

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
c = synthetic.build_config(cfg, 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000)
w = np.zeros(1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, w, c['freqlist'],
        c['band_type'], fr, c['ftype'], c['order'], c['ripple'])
import io, contextlib
for rep in range(3):
    with contextlib.redirect_stdout(io.StringIO()):
        narrow_band_least_squares(*args, rij=c['rij'])
for cold in (True, False):
    acc.clear()
    K = 8
    t0 = time.perf_counter()
    for rep in range(K):
        if cold:
            planner.design_cache_clear()
        with contextlib.redirect_stdout(io.StringIO()):
            out = narrow_band_least_squares(*args, rij=c['rij'])
    tot = (time.perf_counter() - t0) / K
    print('--- %s filter-design cache: whole call %.2f ms, stdict entries %d' % ('cold' if cold else 'warm', tot * 1e3, len(out[4] or {})))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print('  %-28s %7.2f ms' % (k, v / K * 1e3))
