"""Developer: whole calls with the trace resident in HBM (engine.resident_trace) against the same calls with their upload,
interleaved in ONE process (results held until the clock has stopped, as bench.py's timed loop does).
    python tools/ab_resident.py [cfg3] [rounds=6] [calls=12]"""
import contextlib, io, sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import numpy as np
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic, planner, engine

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 12
c = synthetic.build_config(cfg, 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000); w = np.zeros(1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr,
        c['ftype'], c['order'], c['ripple'])
h = engine.get_handle()


def call():
    planner.design_cache_clear()
    with contextlib.redirect_stdout(io.StringIO()):
        return narrow_band_least_squares(*args, rij=c['rij'])


def timed():
    call()
    h.sync()
    held = []
    t = time.perf_counter()
    for _ in range(calls):
        held.append(call())
    dt = (time.perf_counter() - t) / calls * 1e3
    del held
    return dt


for _ in range(5):
    call()
res = {'upload in every call': [], 'trace resident': []}
for r in range(rounds):
    order = list(res) if r % 2 == 0 else list(res)[::-1]
    for k in order:
        if k == 'trace resident':
            with engine.resident_trace(c['st']):
                res[k].append(timed())
        else:
            res[k].append(timed())
for k, v in res.items():
    print('%s: whole call median %.3f ms (rounds: %s)' % (k, np.median(v), ' '.join('%.2f' % x for x in v)), flush=True)
