"""Developer: whole calls on a FRESH copy of the trace every call (a caller that processes another stretch of data each
time) against calls on the same buffers (what bench.py's loop does): the runtime pins the pages of a pageable source on
its first copy and keeps the registration: 2.3 ms for 55 MB of new rows, 1.2 ms for rows it has seen — but most of what a fresh
trace costs a call is the GPU's clock ramp after the idle time in which the caller prepared it (third line: same buffers, same gap).
    python tools/fresh_trace_calls.py [cfg3] [ncalls=12]"""
import contextlib, io, sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import numpy as np
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic, planner, engine

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
c = synthetic.build_config(cfg, 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000); w = np.zeros(1000)
data = np.array([tr.data for tr in c['st']])


def args_for(st):
    return (c['WINLEN_list'], c['overlap'], c['alpha'], st, None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr,
            c['ftype'], c['order'], c['ripple'])


def call(st):
    planner.design_cache_clear()
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        out = narrow_band_least_squares(*args_for(st), rij=c['rij'])
    return (time.perf_counter() - t) * 1e3, out


from narrow_band_least_squares_amd import _hip
up = []
_real = _hip.Handle.upload_rows


def timed_upload(self, rows):
    t = time.perf_counter()
    try:
        return _real(self, rows)
    finally:
        up.append((time.perf_counter() - t) * 1e3)


_hip.Handle.upload_rows = timed_upload
for _ in range(4):
    call(c['st'])
del up[:]
same = [call(c['st'])[0] for _ in range(n)]
print('%s: same buffers every call, back to back: median %.2f ms (upload %.2f ms)' % (cfg, np.median(same), np.median(up)))
del up[:]
gap = []
for _ in range(n):
    time.sleep(0.015)                      # the GPU idles as long as the copy below takes
    gap.append(call(c['st'])[0])
print('   same buffers, 15 ms of idle time before every call: median %.2f ms (upload %.2f ms)' % (np.median(gap), np.median(up)))
keep, fresh = [], []
del up[:]
for _ in range(n):
    st = synthetic.make_stream(data.copy(), c['fs'], starttime=c['st'][0].stats.starttime)       # new buffers, outside the clock
    keep.append(st)                                                                              # (not freed: no address is reused)
    fresh.append(call(st)[0])
print('   a fresh copy of the trace every call (made outside the clock: ~15 ms): median %.2f ms (upload %.2f ms)' % (np.median(fresh), np.median(up)))
