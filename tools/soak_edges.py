"""Developer: ltsva() against the oracle at the edges of the large-array LTS kernel's parameter space: element counts
around the u8 / u16 counter switch (23 / 24 elements = 253 / 276 pairs), the smallest and largest array of that
kernel (9, 32), and trimming fractions from the breakdown point to almost none (h = P - 1).
    python tools/soak_edges.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import contextlib, io
import numpy as np
import nbls_oracle as oracle
import test_gpu_parity as T
from narrow_band_least_squares_amd import synthetic, planner
bad = n = 0
for nchans in (9, 10, 23, 24, 32):
    for alpha in (0.5, 0.51, 0.97, 0.995):
        P = nchans * (nchans - 1) // 2
        fs, winlen = 20.0, 15.0
        npts = int(4.2 * winlen * fs)
        rij = synthetic.array_geometry(nchans, 1.0, seed=900 + nchans)
        data = synthetic.plane_wave(rij, npts, fs, 0.3, 0.4 * fs, baz_deg=37.0 + nchans, vel_kms=0.5, snr_db=3.0,
                                    timing_error_s=0.3, bad_element=nchans - 1, seed=17 * nchans)
        c = dict(fs=fs, rij=rij - rij.mean(axis=1, keepdims=True))
        st = oracle.make_stream(data, fs, starttime=17884.0729166667)
        stf, _, _ = oracle.filter_data(st, 'butter', 0.5, 0.35 * fs, 2, 0.01)
        n += 1
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                T._compare_ltsva(oracle, c, stf, winlen, alpha)
            print('ok   nchans %2d pairs %3d alpha %.3f h %3d' % (nchans, P, alpha, planner.lts_h(P, alpha)), flush=True)
        except Exception as e:      # noqa: BLE001
            bad += 1
            print('FAIL nchans %2d pairs %3d alpha %.3f h %3d: %s %s' % (nchans, P, alpha, planner.lts_h(P, alpha), type(e).__name__, str(e)[:200]), flush=True)
print('edges done: %d cases, %d failures' % (n, bad))
