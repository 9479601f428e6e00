"""Developer: whole calls with the trace resident in HBM (engine.resident_trace) against the same calls with their upload,
interleaved in ONE process (results held until the clock has stopped, as bench.py's timed loop does).
    python tools/ab_resident.py [cfg3] [rounds=6] [calls=12]"""
import contextlib, io, sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import numpy as np
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic, planner, engine

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
nums = [a for a in sys.argv[2:] if a.isdigit()]
rounds = int(nums[0]) if nums else 6
calls = int(nums[1]) if len(nums) > 1 else 12
c = synthetic.build_config(cfg, 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000); w = np.zeros(1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr,
        c['ftype'], c['order'], c['ripple'])
h = engine.get_handle()


def call():
    planner.design_cache_clear()
    with contextlib.redirect_stdout(io.StringIO()):
        return narrow_band_least_squares(*args, rij=c['rij'])


MARK = 'marks' in sys.argv          # where inside the call: execute returned / last batch landed, ms from the call's start
marks = {}
_t0 = [0.0]
_cur = [None]
if MARK:
    from narrow_band_least_squares_amd import _hip
    _ex, _wb = _hip.Handle.execute, _hip.Handle.wait_result_batch

    def ex(self, *a, **k):
        r = _ex(self, *a, **k)
        marks.setdefault((_cur[0], 'execute returned'), []).append((time.perf_counter() - _t0[0]) * 1e3)
        return r

    def wb(self, k_):
        r = _wb(self, k_)
        if k_ == self.result_batches() - 1:
            marks.setdefault((_cur[0], 'last batch landed'), []).append((time.perf_counter() - _t0[0]) * 1e3)
        return r
    _hip.Handle.execute, _hip.Handle.wait_result_batch = ex, wb
PROF = 'prof' in sys.argv          # also the GPU's own time per call (HIP events of the pass; costs a sync per call)
gpu = {}


def timed(key=None):
    call()
    h.sync()
    held = []
    t = time.perf_counter()
    for _ in range(calls):
        _t0[0], _cur[0] = time.perf_counter(), key
        held.append(call())
        marks.setdefault((key, 'returned'), []).append((time.perf_counter() - _t0[0]) * 1e3)
        if PROF:
            gpu.setdefault(key, []).append(h.timings()['total_ms'])
    dt = (time.perf_counter() - t) / calls * 1e3
    del held
    return dt


h.set_profiling(PROF)
for _ in range(5):
    call()
res = {'upload in every call': [], 'trace resident': []}
for r in range(rounds):
    order = list(res) if r % 2 == 0 else list(res)[::-1]
    for k in order:
        if k == 'trace resident':
            with engine.resident_trace(c['st']):
                res[k].append(timed(k))
        else:
            res[k].append(timed(k))
for k, v in res.items():
    print('%s: whole call median %.3f ms (rounds: %s)%s' % (k, np.median(v), ' '.join('%.2f' % x for x in v),
                                                            '; GPU pass inside the calls %.3f ms' % np.median(gpu[k]) if PROF else ''), flush=True)

if MARK:
    for (k, what), v in sorted(marks.items(), key=lambda kv: (str(kv[0][0]), np.median(kv[1]))):
        if k is not None:
            print('   %-22s %-18s median %.3f ms' % (k, what, np.median(v)))
