"""Developer: timeline (ms from the start of the call) of the host-side phases of one whole cfg-3 call."""
import contextlib
import functools
import io
import sys
import time

sys.path.insert(0, '/root/repo')
import numpy as np
from scipy import signal
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic, engine, planner, _hip

events = []
T0 = [0.0]


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    @functools.wraps(f)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            events.append((label, (t - T0[0]) * 1e3, (time.perf_counter() - T0[0]) * 1e3))
    setattr(obj, name, g)


for n in ('prepare', 'launch', 'time_keys', 'time_key_text', 'stdict_from_mask', 'upload_trace', 'stream_rows', 'new_stdict'):
    wrap(engine, n)
nbm = sys.modules['narrow_band_least_squares_amd.narrow_band_least_squares']
wrap(nbm, '_run_bands')
wrap(engine, 'process')
wrap(planner, 'sosfreqz_bands')
wrap(planner, 'design_bandpass_many')
for n in ('plan', 'execute', 'fetch_packed', 'set_trace_from', 'wait_result_batch', 'upload_rows', 'set_geometry', 'set_trace_shape'):
    wrap(_hip.Handle, n, 'handle.' + n)

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
c = synthetic.build_config(cfg, 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000)
w = np.zeros(1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, w, c['freqlist'],
        c['band_type'], fr, c['ftype'], c['order'], c['ripple'])
HOLD = 'hold' in sys.argv[2:]     # keep every call's result alive (what bench.py's timed loop does)
RESIDENT = 'resident' in sys.argv[2:]     # the calls run inside `with engine.resident_trace(st)`: no upload
kept = []
ctx = engine.resident_trace(c['st']) if RESIDENT else contextlib.nullcontext()
ctx.__enter__()
allev, totals = [], []
for rep in range(16 if HOLD else 6):
    planner.design_cache_clear()
    events.clear()
    T0[0] = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        out = narrow_band_least_squares(*args, rij=c['rij'])
    total = (time.perf_counter() - T0[0]) * 1e3
    totals.append(total)
    allev.append(sorted(events, key=lambda e: e[1]))
    if HOLD:
        kept.append(out)
    del out
ctx.__exit__(None, None, None)
print('whole call %.2f ms%s (last of %d; median of the last %d: %.2f)' % (total, ' (trace resident)' if RESIDENT else '', len(totals), len(totals) - 4, np.median(totals[4:])))
for label, a, b in allev[-1]:
    print('  %6.2f .. %6.2f  (%5.2f)  %s' % (a, b, b - a, label))
# the same events over the calls after the first four (same sequence of labels in every call): medians
seq = [e[0] for e in allev[-1]]
rest = [ev for ev in allev[4:] if [e[0] for e in ev] == seq]
if len(rest) > 2:
    print('medians over %d calls:' % len(rest))
    for i, label in enumerate(seq):
        a = np.median([ev[i][1] for ev in rest]); b = np.median([ev[i][2] for ev in rest])
        print('  %6.2f .. %6.2f  (%5.2f)  %s' % (a, b, np.median([ev[i][2] - ev[i][1] for ev in rest]), label))
