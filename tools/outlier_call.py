"""Developer: one call in the first ~20 of a process takes 5-10 ms longer than its neighbours — where?  Per-call times of the
handle's entry points; prints the slowest call's breakdown beside the median call's.    python tools/outlier_call.py [cfg1b] [ncalls=40]"""
import contextlib, functools, io, sys, time, gc
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import numpy as np
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic, planner, engine, _hip

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg1b'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
acc = {}


def wrap(obj, name, label):
    f = getattr(obj, name)

    @functools.wraps(f)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t) * 1e3
    setattr(obj, name, g)


for m in ('plan', 'execute', 'fetch_packed', 'wait_result_batch', 'upload_rows', 'set_geometry', 'set_trace_shape', 'stream_results', 'set_uncertainty', 'reserve_results', 'sync'):
    wrap(_hip.Handle, m, 'handle.' + m)
for m in ('prepare', 'stdict_from_mask', 'time_key_text', 'release_deferred'):
    wrap(engine, m, m)
wrap(planner, 'sosfreqz_bands', 'sosfreqz_bands')
c = synthetic.build_config(cfg, 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000); w = np.zeros(1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr,
        c['ftype'], c['order'], c['ripple'])
rows = []
for i in range(n):
    planner.design_cache_clear()
    acc.clear()
    g0 = gc.get_count()
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        out = narrow_band_least_squares(*args, rij=c['rij'])
    rows.append(((time.perf_counter() - t) * 1e3, dict(acc), g0, gc.get_count()))
    del out
tot = [r[0] for r in rows]
print(' '.join('%.1f' % x for x in tot))
k = int(np.argmax(tot[3:])) + 3
med = int(np.argsort(tot)[len(tot) // 2])
for name, i in (('slowest (call %d)' % k, k), ('median (call %d)' % med, med)):
    print('%s: %.2f ms; gc counts %s -> %s' % (name, rows[i][0], rows[i][2], rows[i][3]))
    for lab, v in sorted(rows[i][1].items(), key=lambda kv: -kv[1]):
        print('    %-26s %.3f' % (lab, v))
