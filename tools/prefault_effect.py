"""Developer (r04): does a pre-faulted, retained malloc heap remove the slow first ~200 calls of a process?  (The whole call
allocates ~12 MB of fresh memory per call while its results are kept alive: page faults, some of them contending with the
upload thread's page pinning.)    python tools/prefault_effect.py [0|1]"""
import contextlib, ctypes, io, sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if mode:
    libc = ctypes.CDLL('libc.so.6')
    M_TRIM_THRESHOLD, M_MMAP_THRESHOLD, M_TOP_PAD = -1, -3, -2
    libc.mallopt(M_MMAP_THRESHOLD, 1 << 30)      # big blocks come from the heap, not from mmap
    libc.mallopt(M_TRIM_THRESHOLD, -1 if False else (1 << 31) - 1)   # ... and freed heap is kept
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    n = 768 << 20
    p = libc.malloc(n)
    ctypes.memset(p, 1, n)                        # fault the pages in
    libc.free(p)
import numpy as np
from narrow_band_least_squares_amd import narrow_band_least_squares, synthetic, planner
c = synthetic.build_config('cfg3', 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000); w = np.zeros(1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr, c['ftype'], c['order'], c['ripple'])
def call():
    planner.design_cache_clear()
    with contextlib.redirect_stdout(io.StringIO()):
        return narrow_band_least_squares(*args, rij=c['rij'])
for _ in range(5):
    call()
held, ts = [], []
for rep in range(20):
    t = time.perf_counter(); held.append(call()); ts.append((time.perf_counter() - t) * 1e3)
print('prefault=%d: 5 warm-up + 20 held calls: mean %.2f median %.2f min %.2f max %.2f ms' % (mode, np.mean(ts), np.median(ts), min(ts), max(ts)))
