"""GPU ``ltsva``: drop-in for ``from lts_array import ltsva`` (reference call sites
narrow_band_least_squares.py:91,183 and example.py:109; the lts_array source itself is an
empty git submodule in the reference checkout, so the algorithm follows the published one as
summarised in SURVEY.md §3.3).
"""
import numpy as np

from . import engine
from .helpers import get_rij


def ltsva(st, lat_list, lon_list, window_length, window_overlap, alpha=1.0,
          plot_array_coordinates=False, rij=None):
    """Window the (already filtered) stream, pick pairwise cross-correlation lags and solve for
    the slowness vector by OLS (``alpha == 1.0``) or FAST-LTS (``0.5 <= alpha < 1``).

    Returns ``(vel, baz, t, mdccm, stdict, sigma_tau, conf_int_vel, conf_int_baz)``:
    trace velocity km/s, back-azimuth degrees in [0, 360), window-centre times as matplotlib
    date numbers, median cross-correlation maximum, dropped-element dictionary (``{}`` for
    OLS), sigma_tau seconds, and the 90 % confidence half-widths of trace velocity (km/s) and
    back-azimuth (degrees; NaN where the direction is undetermined).  ``rij`` (2, N) km overrides the lat/lon geometry."""
    data, fs, t0 = engine.stream_rows(st)
    nchans = len(data)
    engine.check_elements(nchans, alpha)
    if rij is None:
        rij = get_rij(lat_list, lon_list, nchans)
    if alpha == 1.0:
        print('ALPHA is 1.0. Performing an ordinary least squares fit, NOT least trimmed squares.')
    if plot_array_coordinates:
        import warnings
        warnings.warn('plot_array_coordinates is not supported on the HIP path; ignored.')
    def host_side(res):        # key text of the dropped-element dictionary: needs no GPU result
        if alpha < 1.0:
            res.keys = engine.time_keys(res.t, res.nwin)

    res = engine.process(data, fs, t0, rij, [(None, None)], [window_length], window_overlap, alpha,
                         prefiltered=True, host_overlap=host_side, want_uncert=True)
    n = int(res.nwin[0])
    vel = res.vel[0, :n].copy()
    baz = res.baz[0, :n].copy()
    t = res.t[0, :n].copy()
    mdccm = res.mdccm[0, :n].copy()
    sigma_tau = res.sigma_tau[0, :n].copy()
    if alpha == 1.0:
        stdict = {}
    else:
        stdict = engine.stdict_from_mask(res.mask, res.nwin, res.pair_idx, nchans, res.keys)
    # (Szuberla & Olson confidence intervals: per-unit scalar math on the GPU behind the solve, csrc/solve.hip:
    #  uncertainty_kernel — no host loop over windows)
    conf_int_vel = res.vel_uncert[0, :n].copy()
    conf_int_baz = res.baz_uncert[0, :n].copy()
    return vel, baz, t, mdccm, stdict, sigma_tau, conf_int_vel, conf_int_baz
