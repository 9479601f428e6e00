#!/usr/bin/env python3
"""Offline counterpart of the reference's example.py: the same sequence of calls with the same positional
arguments (filter_data -> ltsva broadband; get_freqlist / get_winlenlist -> narrow_band_least_squares ->
write_txtfile), on a synthetic 8-element plane wave instead of an IRIS download (there is no network here), and
without the matplotlib figures.  Needs an MI355X.

    python examples/synthetic_example.py [--alpha 0.5] [--parallel]

The three import lines are the only difference from a script written against the reference: they name this
package instead of `narrow_band_least_squares`, `helpers` and `lts_array`
(or call narrow_band_least_squares_amd.install_as_reference_modules() and keep the original imports).
"""
import argparse
import math
import os
import sys
import tempfile

import numpy as np
from scipy import signal

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from narrow_band_least_squares_amd import ltsva                                              # noqa: E402
from narrow_band_least_squares_amd import narrow_band_least_squares, narrow_band_least_squares_parallel  # noqa: E402
from narrow_band_least_squares_amd import (get_freqlist, get_winlenlist, filter_data, get_rij, write_txtfile,  # noqa: E402
                                           read_txtfile, synthetic)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--alpha', type=float, default=1.0, help='1.0 = ordinary least squares, < 1 = least trimmed squares')
    ap.add_argument('--parallel', action='store_true', help='use narrow_band_least_squares_parallel (all visible GPUs)')
    args = ap.parse_args()

    # ---- user input, as in example.py ----
    FMIN, FMAX, NBANDS = 0.1, 5.0, 8
    FREQ_BAND_TYPE, FILTER_TYPE, FILTER_ORDER, FILTER_RIPPLE = 'log', 'cheby1', 2, 0.01
    WINOVER, WINDOW_LENGTH_TYPE, WINLEN, WINLEN_1, WINLEN_X = 0.5, 'adaptive', 50, 60, 30
    ALPHA, PLOT_ARRAY_COORDINATES = args.alpha, False

    # ---- data: 20 minutes at 20 Hz on 8 elements, plane wave from 225 deg at 0.34 km/s, one mistimed element ----
    rij0 = synthetic.array_geometry(8, 1.0)
    data = synthetic.plane_wave(rij0, 24001, 20.0, FMIN, FMAX, timing_error_s=0.25 if ALPHA < 1.0 else 0.0,
                                bad_element=7 if ALPHA < 1.0 else None)
    latlist, lonlist = synthetic.latlon_from_rij(rij0)
    st = synthetic.make_stream(data, 20.0, lat=latlist, lon=lonlist)
    nchans = len(st)
    rij = get_rij(latlist, lonlist, nchans)
    print('array aperture %.2f km, %d elements' % (np.ptp(rij, axis=1).max(), nchans))

    # ---- broadband least squares ----
    stf_broad, Fs, sos = filter_data(st, FILTER_TYPE, FMIN, FMAX, FILTER_ORDER, FILTER_RIPPLE)
    (vel_broad, baz_broad, t_broad, mdccm_broad, stdict_broad, sig_tau_broad, vel_uncert_broad,
     baz_uncert_broad) = ltsva(stf_broad, latlist, lonlist, WINLEN, WINOVER, ALPHA, PLOT_ARRAY_COORDINATES)
    print('broadband: %d windows, median back-azimuth %.1f deg, trace velocity %.3f km/s, MdCCM %.2f'
          % (len(vel_broad), np.median(baz_broad), np.median(vel_broad), np.median(mdccm_broad)))
    freq_resp_list = np.logspace(math.log(0.01, 10), math.log(Fs / 2, 10), num=1000)
    w_broad, h_broad = signal.sosfreqz(sos, freq_resp_list, fs=Fs)

    # ---- narrow-band least squares ----
    freqlist, NBANDS, FMAX = get_freqlist(FMIN, FMAX, FREQ_BAND_TYPE, NBANDS)
    WINLEN_list = get_winlenlist(WINDOW_LENGTH_TYPE, NBANDS, WINLEN, WINLEN_1, WINLEN_X)
    run = narrow_band_least_squares_parallel if args.parallel else narrow_band_least_squares
    (vel_array, baz_array, mdccm_array, t_array, stdict_all, sig_tau_array, num_compute_list, w_array,
     h_array) = run(WINLEN_list, WINOVER, ALPHA, st, latlist, lonlist, NBANDS, w_broad, h_broad, freqlist, FREQ_BAND_TYPE,
                    freq_resp_list, FILTER_TYPE, FILTER_ORDER, FILTER_RIPPLE)
    for b in range(NBANDS):
        n = num_compute_list[b]
        print('band %d  %.3f-%.3f Hz  %3d windows  baz %.1f  vel %.3f  MdCCM %.2f'
              % (b + 1, freqlist[b], freqlist[b + 1], n, np.median(baz_array[b, :n]), np.median(vel_array[b, :n]),
                 np.median(mdccm_array[b, :n])))
    if stdict_all is not None:
        dropped = np.concatenate([v for k, v in stdict_all.items() if k != 'size'])
        print('LTS dropped element pairs in %d windows; element flagged most often: %d'
              % (len(stdict_all) - 1, np.bincount(dropped).argmax()))

    # ---- the reference's text format ----
    d = tempfile.mkdtemp() + '/'
    write_txtfile(d, 'narrow_band_results', vel_array, baz_array, mdccm_array, t_array, freqlist, num_compute_list)
    back = read_txtfile(d, 'narrow_band_results')
    print('wrote and re-read %s: %d bands' % (d + 'narrow_band_results.txt', back[6]))


if __name__ == '__main__':
    main()
