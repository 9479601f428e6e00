"""Worker for the world_size-2 gloo test: runs narrow_band_least_squares_parallel() on every rank
with the device pass replaced by an oracle-backed stand-in (there is no GPU in the CPU test
environment), so that what is tested is the host logic of the N>1 path: band partition, padded
all-gather, reassembly, stdict merge."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import nbls_oracle as oracle  # noqa: E402
from narrow_band_least_squares_amd import engine, planner  # noqa: E402
import importlib  # noqa: E402
nbls_mod = importlib.import_module('narrow_band_least_squares_amd.narrow_band_least_squares')


def fake_process(data, fs, t0, rij, band_edges, winlens, winover, alpha, filter_type=None, filter_order=None,
                 filter_ripple=None, vector_len=None, window_slice=None, **kw):
    nb = len(band_edges)
    nchans = data.shape[0]
    xij, pair_idx, _ = planner.co_array(rij)
    P = xij.shape[0]
    vel = np.zeros((nb, vector_len)); baz = np.zeros_like(vel); md = np.zeros_like(vel)
    sig = np.zeros_like(vel); t = np.zeros_like(vel)
    wts = np.zeros((nb, vector_len, P), dtype=np.uint8)
    nwin = np.zeros(nb, dtype=int)
    sos = []
    for b, (fmin, fmax) in enumerate(band_edges):
        st = oracle.make_stream(data, fs, starttime=t0)
        stf, _, s = oracle.filter_data(st, filter_type, fmin, fmax, filter_order, filter_ripple)
        sos.append(s)
        out, internals = oracle.ltsva(stf, None, None, winlens[b], winover, alpha, rij=rij, return_internals=True)
        n = len(out[0])
        nwin[b] = n
        vel[b, :n], baz[b, :n], t[b, :n], md[b, :n], sig[b, :n] = out[0], out[1], out[2], out[3], out[5]
        wts[b, :n] = internals['weights'].T
        if window_slice is not None:          # keep only this rank's slice of the rows, like the device path
            k, nsl = window_slice
            lo, hi = (n * k) // nsl, (n * (k + 1)) // nsl
            for arr in (vel, baz, md, sig):
                arr[b, :lo] = 0.0
                arr[b, hi:] = 0.0
            wts[b, :lo] = 0
            wts[b, hi:] = 0
    return engine.BandBatch(vel=vel, baz=baz, mdccm=md, sigma_tau=sig, nwin=nwin, t=t, weights=wts, sos=sos,
                            pair_idx=pair_idx, nchans=nchans)


def main():
    import torch.distributed as td
    td.init_process_group('gloo')
    rank = td.get_rank()
    engine.process = fake_process
    if len(sys.argv) > 2:
        os.environ['NBLS_SHARD'] = sys.argv[2]
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', sys.argv[1] + '.npz'), allow_pickle=False)
    st = oracle.make_stream(gold['data'], float(gold['fs']), starttime=17884.0729166667)
    nb = len(gold['num_compute'])
    fr = gold['freq_resp']
    w = np.zeros(len(fr))
    args = (list(gold['winlens']), 0.5, float(gold['alpha']), st, None, None, nb, w, w, list(gold['freqlist']),
            str(gold['band_type']), fr, str(gold['ftype']), 2, 0.01)
    got = nbls_mod.narrow_band_least_squares_parallel(*args, rij=gold['rij'])
    exp = oracle.narrow_band_least_squares(*args, rij=gold['rij'])
    assert got[6] == exp[6], (got[6], exp[6])
    for i in (0, 1, 2, 3, 5, 7, 8):
        np.testing.assert_array_equal(got[i], exp[i])
    if exp[4] is None:
        assert got[4] is None
    else:
        assert list(got[4].keys()) == list(exp[4].keys())
        for k in exp[4]:
            np.testing.assert_array_equal(got[4][k], exp[4][k])
    td.barrier()
    if rank == 0:
        print('DIST_OK world=%d bands=%d' % (td.get_world_size(), nb))
    td.destroy_process_group()


if __name__ == '__main__':
    main()
