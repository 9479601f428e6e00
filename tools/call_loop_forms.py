"""Developer: the whole cfg-3 call in the three loop forms a caller (or a benchmark) may use — results kept alive, results
dropped outside the clock, results replaced by the next call's (the previous dictionary is torn down INSIDE the next call's
time: what `for ...: out = narrow_band_least_squares(...)` pays)."""
import contextlib, io, sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import numpy as np
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic, planner
c = synthetic.build_config('cfg3', 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000); w = np.zeros(1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr, c['ftype'], c['order'], c['ripple'])
def call():
    planner.design_cache_clear()
    with contextlib.redirect_stdout(io.StringIO()):
        return narrow_band_least_squares(*args, rij=c['rij'])
for _ in range(4):
    call()
import queue, threading
for form in ('held', 'dropped outside the clock', 'replaced by the next call', 'handed to a consumer thread'):
    ts, held, out = [], [], None
    q = queue.Queue()
    def reaper():
        while True:
            item = q.get()
            if item is None:
                return
            del item
    th = threading.Thread(target=reaper)
    th.start()
    t_all = time.perf_counter()
    for rep in range(24):
        t = time.perf_counter()
        if form == 'held':
            held.append(call())
        elif form.startswith('dropped'):
            out = call()
        elif form.startswith('handed'):
            q.put(call())         # the consumer releases it while the next call waits for its GPU batches
        else:
            out = call()          # rebinding frees the previous result here, inside the clock
        ts.append((time.perf_counter() - t) * 1e3)
        if form.startswith('dropped'):
            out = None
    q.put(None)
    th.join()
    t_all = (time.perf_counter() - t_all) / 24 * 1e3
    del held
    print('%-28s median %.2f  min %.2f  max %.2f ms; loop mean incl. everything %.2f ms' % (form, np.median(ts[4:]), min(ts[4:]), max(ts[4:]), t_all))
# what the teardown of ONE result costs, piece by piece
import gc
out = call()
t = time.perf_counter(); d = out[4]; n = len(d); vals = list(d.values()); keys = list(d.keys()); t1 = time.perf_counter()
del out
t2 = time.perf_counter(); del keys; t3 = time.perf_counter(); d.clear(); t4 = time.perf_counter(); del vals; t5 = time.perf_counter()
from narrow_band_least_squares_amd import engine
engine.release_deferred(); t6 = time.perf_counter(); gc.collect(); t7 = time.perf_counter()
print('teardown of one result (%d entries): tuple+grids %.2f ms, key list (strings survive in the dict) %.2f, dict.clear (keys die) %.2f, value list (arrays die) %.2f, deferred helpers %.2f, gc.collect %.2f'
      % (n, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3, (t7 - t6) * 1e3))
