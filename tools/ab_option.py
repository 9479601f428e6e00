"""Developer: A/B of one library option on whole calls, interleaved in ONE process (boxes differ by more than most options do).
    python tools/ab_option.py result_tail_units -1 0 1024 4096 [cfg3] [rounds=6] [calls=12]
    python tools/ab_option.py py:ROW_PIPELINE 0 1            (a module switch of engine.py instead of a library option)
Every round runs `calls` whole calls per value (results held until the clock has stopped, as bench.py's timed loop does);
prints the median over rounds of the per-round mean."""
import contextlib, io, sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import numpy as np
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic, planner, engine

key = sys.argv[1]
vals, rest = [], []
for a in sys.argv[2:]:
    (vals if a.lstrip('-').isdigit() and not rest else rest).append(a)
vals = [int(v) for v in vals]
cfg = rest[0] if rest else 'cfg3'
rounds = int(rest[1]) if len(rest) > 1 else 6
calls = int(rest[2]) if len(rest) > 2 else 12
share = None
if ':' in cfg:                      # cfg4:0/8 = the k-th of n band shares (dist.shard_bands), as bench.py runs cfg-4 on one GPU
    cfg, share = cfg.split(':')
c = synthetic.build_config(cfg, 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000); w = np.zeros(1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr,
        c['ftype'], c['order'], c['ripple'])
h = engine.get_handle()
if share:
    from narrow_band_least_squares_amd import dist
    k, n = (int(x) for x in share.split('/'))
    rows, fs_, t0_ = engine.stream_rows(c['st'])
    costs = dist.band_costs(c['npts'], c['fs'], list(c['WINLEN_list']), c['overlap'], c['N'] * (c['N'] - 1) // 2)
    bands = dist.shard_bands(costs, n)[k]
    edges = [(c['freqlist'][b], c['freqlist'][b + 1]) for b in bands]
    wl = [c['WINLEN_list'][b] for b in bands]


def call():
    planner.design_cache_clear()
    with contextlib.redirect_stdout(io.StringIO()):
        if share:
            return engine.process(rows, fs_, t0_, c['rij'], edges, wl, c['overlap'], c['alpha'], c['ftype'], c['order'], c['ripple'])
        return narrow_band_least_squares(*args, rij=c['rij'])


for _ in range(5):
    call()
res = {v: [] for v in vals}
for r in range(rounds):
    for v in (vals if r % 2 == 0 else vals[::-1]):
        if key.startswith('py:'):
            setattr(engine, key[3:], type(getattr(engine, key[3:]))(v))     # a module switch of engine.py (py:ROW_PIPELINE 0 1)
        else:
            h.set_option(key, v)
        call()
        h.sync()
        held = []
        t = time.perf_counter()
        for _ in range(calls):
            held.append(call())
        res[v].append((time.perf_counter() - t) / calls * 1e3)
        del held
for v in vals:
    print('%s = %d: whole call median %.3f ms (rounds: %s)' % (key, v, np.median(res[v]), ' '.join('%.2f' % x for x in res[v])), flush=True)
