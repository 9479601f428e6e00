"""Trace-velocity and back-azimuth confidence intervals of the slowness estimate — the 7th and 8th
returns of ``ltsva`` (``vel_uncert``, ``baz_uncert`` at narrow_band_least_squares.py:91; they are
discarded by the narrow-band loop but returned to broadband callers, example.py:109).

Szuberla & Olson (2004) as used by lts_array [R: source absent from the reference checkout]: the
90 % confidence region of the slowness vector is an ellipse centred on the estimate z, with semi-axes
``sqrt(chi2_{0.90,2}) * sigma_tau / sqrt(lambda_i)`` along the eigenvectors of X^T X (X = co-array).
``conf_int_vel`` is half the spread of 1/|s| over the ellipse, ``conf_int_baz`` half the angle the
ellipse subtends at the origin (NaN when the origin is inside, i.e. the direction is undetermined).
Per-window scalar math on the host, vectorised over windows (the outputs ride on z and sigma_tau
that the solve kernel already returns).
"""
import numpy as np

CHI2_90_2DOF = -2.0 * np.log(1.0 - 0.90)          # chi2.ppf(0.90, 2) = 4.605170185988092


def confidence_intervals(xij, z, sigma_tau, nsample=720, newton_iters=8):
    """xij (P, 2) km; z (n, 2) s/km; sigma_tau (n,) s -> (conf_int_vel (n,) km/s, conf_int_baz (n,) deg)."""
    z = np.asarray(z, dtype=np.float64).reshape(-1, 2)
    sig = np.asarray(sigma_tau, dtype=np.float64).reshape(-1)
    n = len(sig)
    ci_vel = np.full(n, np.nan)
    ci_baz = np.full(n, np.nan)
    if n == 0:
        return ci_vel, ci_baz
    evals, evecs = np.linalg.eigh(xij.T @ xij)
    ang = np.arccos(np.clip(evecs[0, 0], -1.0, 1.0))
    R = np.array([[np.cos(ang), np.sin(ang)], [-np.sin(ang), np.cos(ang)]])
    with np.errstate(invalid='ignore', divide='ignore'):
        a = np.sqrt(CHI2_90_2DOF) * sig / np.sqrt(evals[0])
        b = np.sqrt(CHI2_90_2DOF) * sig / np.sqrt(evals[1])
        c = z @ R.T                                   # ellipse centre in the eigen-frame
        x0, y0 = c[:, 0], c[:, 1]
        ok = np.isfinite(a) & np.isfinite(b) & np.isfinite(x0) & np.isfinite(y0)

        # ---- radial extrema: stationary points of f(phi) = |c + (a cos phi, b sin phi)|^2 ----
        phi = np.linspace(0.0, 2.0 * np.pi, nsample, endpoint=False)
        cx = x0[:, None] + a[:, None] * np.cos(phi)[None, :]
        cy = y0[:, None] + b[:, None] * np.sin(phi)[None, :]
        f = cx * cx + cy * cy
        r_ext = []
        for pick in (np.nanargmin, np.nanargmax):
            f_safe = np.where(np.isfinite(f), f, 0.0)
            p = phi[pick(f_safe, axis=1)]
            for _ in range(newton_iters):
                s, co = np.sin(p), np.cos(p)
                g = -a * s * (x0 + a * co) + b * co * (y0 + b * s)                 # f'/2
                h = -a * co * x0 - a * a * (co * co - s * s) - b * s * y0 + b * b * (co * co - s * s)   # f''/2
                step = np.where(np.abs(h) > 0, g / h, 0.0)
                p = p - np.clip(step, -0.05, 0.05)
            r_ext.append(np.hypot(x0 + a * np.cos(p), y0 + b * np.sin(p)))
        rmin, rmax = r_ext
        ci_vel = 0.5 * np.abs(1.0 / rmin - 1.0 / rmax)

        # ---- subtended angle: tangents from the origin (unit circle after scaling by the semi-axes) ----
        pz = np.where(a > 0, x0 / a, np.inf)
        qz = np.where(b > 0, y0 / b, np.inf)
        d2 = pz * pz + qz * qz
        outside = d2 > 1.0
        root = np.sqrt(np.where(outside, d2 - 1.0, np.nan)) / d2
        k = 1.0 - 1.0 / d2
        t1 = np.stack((a * (pz * k - root * qz), b * (qz * k + root * pz)), axis=1)
        t2 = np.stack((a * (pz * k + root * qz), b * (qz * k - root * pz)), axis=1)
        t1 = t1 @ R                                   # back to the east/north frame
        t2 = t2 @ R
        th1 = (np.degrees(np.arctan2(t1[:, 1], t1[:, 0])) - 360.0) % 360.0
        th2 = (np.degrees(np.arctan2(t2[:, 1], t2[:, 0])) - 360.0) % 360.0
        dth = np.abs(th1 - th2)
        dth = np.where(dth > 180.0, np.abs(dth - 360.0), dth)
        ci_baz = 0.5 * dth
        zero = ok & (a == 0) & (b == 0)               # exact fit: a point, no spread
        ci_vel = np.where(zero, 0.0, ci_vel)
        ci_baz = np.where(zero, 0.0, ci_baz)
        ci_baz = np.where(ok & ~zero & ~outside, np.nan, ci_baz)
    ci_vel = np.where(ok, ci_vel, np.nan)
    ci_baz = np.where(ok, ci_baz, np.nan)
    return ci_vel, ci_baz
