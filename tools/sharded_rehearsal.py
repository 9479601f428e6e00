"""Developer: the band-sharded call with N ranks of ONE process on this box's one GPU over the tests' loopback transport
(tests/c_caller/libloopback_rccl.so) — how long does the host work for AFTER the gather has delivered?  (r03: the whole
dictionary, 7.5 ms at cfg-3; r04: the dictionary is built from the ranks' streamed batches while the GPUs work.)  The
GPU times mean nothing here (N ranks share one GPU); the host terms do.

    python tools/sharded_rehearsal.py [nranks=8] [cfg3]            NBLS_STREAM_RESULTS=0: the r03 behaviour
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/sharded_rehearsal.py N
        the launcher form (one process per rank, all on GPU 0): every rank streams ITS share's entries, the others' follow the gather"""
import contextlib, io, os, sys, time
ROOT = __file__.rsplit('/', 2)[0]
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
launcher = int(os.environ.get('WORLD_SIZE', '1')) > 1       # python -m torch.distributed.run --nproc-per-node N tools/sharded_rehearsal.py N
if launcher:
    n = int(os.environ['WORLD_SIZE'])
    os.environ['NBLS_DEVICE'] = '0'
else:
    os.environ['NBLS_DEVICES'] = ','.join(['0'] * n)
os.environ['NBLS_FORCE_DIST_PATH'] = '1'
import numpy as np
from narrow_band_least_squares_amd import dist, engine, planner, synthetic, narrow_band_least_squares_parallel, narrow_band_least_squares
dist.set_transport_library(os.path.join(ROOT, 'tests', 'c_caller', 'libloopback_rccl.so'), allow_shared_device=True)
c = synthetic.build_config(sys.argv[2] if len(sys.argv) > 2 else 'cfg3', 1.0)
fr = np.logspace(-2, np.log10(c['fs'] / 2), 1000); w = np.zeros(1000)
args = (c['WINLEN_list'], c['overlap'], c['alpha'], c['st'], None, None, c['NBANDS'], w, w, c['freqlist'], c['band_type'], fr, c['ftype'], c['order'], c['ripple'])
marks = {}
real_gather = dist.Group.gather
def gather(self, *a, **k):
    marks['gather_in'] = time.perf_counter()
    out = real_gather(self, *a, **k)
    marks['gather_out'] = time.perf_counter()
    return out
dist.Group.gather = gather
tot, post, pre = [], [], []
for rep in range(12):
    planner.design_cache_clear()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        out = narrow_band_least_squares_parallel(*args, rij=c['rij'])
    t1 = time.perf_counter()
    tot.append((t1 - t0) * 1e3); post.append((t1 - marks['gather_out']) * 1e3); pre.append((marks['gather_in'] - t0) * 1e3)
with contextlib.redirect_stdout(io.StringIO()):
    ser = narrow_band_least_squares(*args, rij=c['rij'])
for i in (0, 1, 2, 3, 5, 7, 8):
    assert np.array_equal(out[i], ser[i])
assert list(out[4].keys()) == list(ser[4].keys())
print('%s%d ranks on one GPU, stream=%s: whole call median %.2f ms; until the gather is entered %.2f ms; AFTER the gather has delivered %.2f ms (dictionary: %d entries); equal to the serial call'
      % ('launcher form, rank %s of ' % os.environ['RANK'] if launcher else '', n, os.environ.get('NBLS_STREAM_RESULTS', '1'), np.median(tot[3:]), np.median(pre[3:]), np.median(post[3:]), len(out[4])))
