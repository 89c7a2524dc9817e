"""
oracle/qd_oracle/numerics.py -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).

Pure-NumPy restatements of the third-party array primitives the reference's
per-timestep path leans on (numpy 2.2 / scipy.ndimage 1.15 semantics as
measured in SURVEY.md Appendix B).  Nothing here imports scipy, so the oracle
runs on a box that only has numpy.

Each function cites the reference call site whose behaviour it pins.
"""
from __future__ import annotations

import numpy as np

DBL_MAX = np.finfo(np.float64).max


def nan_to_num(x):
    """np.nan_to_num: NaN->0, +inf->DBL_MAX, -inf->-DBL_MAX (dynamics.py:661-667)."""
    return np.nan_to_num(x)


def gradient_axis0(F, d):
    """np.gradient(F, d, axis=0) (dynamics.py:167-168,489): centred interior
    ``(F[i+1]-F[i-1])/(2 d)``, first-order one-sided at both edges."""
    out = np.empty_like(F)
    out[1:-1] = (F[2:] - F[:-2]) / (2.0 * d)
    out[0] = (F[1] - F[0]) / d
    out[-1] = (F[-1] - F[-2]) / d
    return out


def gradient_axis1(F, d):
    """np.gradient(F, d, axis=1) (dynamics.py:488): NOT periodic in longitude."""
    out = np.empty_like(F)
    out[:, 1:-1] = (F[:, 2:] - F[:, :-2]) / (2.0 * d)
    out[:, 0] = (F[:, 1] - F[:, 0]) / d
    out[:, -1] = (F[:, -1] - F[:, -2]) / d
    return out


def wrap_coord(x, n):
    """Coordinate folding of scipy.ndimage.map_coordinates(mode='wrap')
    (dynamics.py:117, ocean.py:193, run_simulation.py:1157).  Period is n-1:
      x < 0    -> x + (n-1) * (trunc(-x/(n-1)) + 1)
      x > n-1  -> x - (n-1) * trunc(x/(n-1))
    values already inside [0, n-1] are used as they are."""
    x = np.asarray(x, dtype=np.float64)
    sz = float(n - 1)
    out = x.copy()
    if n <= 1:
        out[...] = 0.0
        return out
    neg = x < 0.0
    if np.any(neg):
        xn = x[neg]
        out[neg] = xn + sz * (np.trunc(-xn / sz) + 1.0)
    big = x > sz
    if np.any(big):
        xb = x[big]
        out[big] = xb - sz * np.trunc(xb / sz)
    return out


def bilinear_wrap(field, dep_row, dep_col):
    """map_coordinates(field, [dep_row, dep_col], order=1, mode='wrap',
    prefilter=False).  Corner accumulation order follows scipy's
    NI_GeometricTransform: t = 0; t += f00*wr0*wc0; t += f01*wr0*wc1;
    t += f10*wr1*wc0; t += f11*wr1*wc1."""
    nr, nc = field.shape
    dep_row = np.asarray(dep_row, dtype=np.float64)
    dep_col = np.asarray(dep_col, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        r = np.nan_to_num(wrap_coord(dep_row, nr), nan=0.0, posinf=0.0, neginf=0.0)
        c = np.nan_to_num(wrap_coord(dep_col, nc), nan=0.0, posinf=0.0, neginf=0.0)
    r0f = np.floor(r)
    c0f = np.floor(c)
    tr = r - r0f
    tc = c - c0f
    r0 = r0f.astype(np.int64)
    c0 = c0f.astype(np.int64)
    # neighbour index can only leave the array when its weight is exactly 0
    r0 = np.clip(r0, 0, nr - 1)
    c0 = np.clip(c0, 0, nc - 1)
    r1 = np.minimum(r0 + 1, nr - 1)
    c1 = np.minimum(c0 + 1, nc - 1)
    wr0 = 1.0 - tr
    wr1 = tr
    wc0 = 1.0 - tc
    wc1 = tc
    t = np.zeros(r.shape, dtype=np.float64)
    t = t + field[r0, c0] * wr0 * wc0
    t = t + field[r0, c1] * wr0 * wc1
    t = t + field[r1, c0] * wr1 * wc0
    t = t + field[r1, c1] * wr1 * wc1
    # a NaN coordinate is "outside" for scipy: the constant fill value 0.0 (measured with scipy 1.15.3; infinite or
    # astronomically large coordinates hit an undefined float->int cast inside scipy and are not pinned)
    bad = np.isnan(dep_row) | np.isnan(dep_col)
    if np.any(bad):
        t = np.where(bad, 0.0, t)
    return t


def _take(F, idx, axis):
    return np.take(F, idx, axis=axis)


def _edge_index(i, n, mode):
    """Index extension for scipy.ndimage filters (convolve / gaussian_filter):
    'wrap' = true period n, 'nearest' = clamp, 'reflect' = d c b a | a b c d."""
    if mode == "wrap":
        return np.mod(i, n)
    if mode == "nearest":
        return np.clip(i, 0, n - 1)
    if mode == "reflect":
        p = 2 * n
        j = np.mod(i, p)
        return np.where(j >= n, p - 1 - j, j)
    raise ValueError(mode)


def correlate1d_sym(F, w, axis, mode):
    """scipy.ndimage.correlate1d with a symmetric odd kernel (what
    gaussian_filter1d calls).  Accumulation order of NI_Correlate1D's symmetric
    branch: tmp = x[0]*w0; for j=-r..-1: tmp += (x[j] + x[-j]) * w[j]."""
    n = F.shape[axis]
    r = (len(w) - 1) // 2
    base = np.arange(n)
    out = _take(F, base, axis) * w[r]
    for j in range(-r, 0):
        lo = _take(F, _edge_index(base + j, n, mode), axis)
        hi = _take(F, _edge_index(base - j, n, mode), axis)
        out = out + (lo + hi) * w[r + j]
    return out


def gaussian_kernel1d(sigma, truncate=4.0):
    """scipy.ndimage._gaussian_kernel1d(sigma, 0, radius) with
    radius=int(truncate*sigma+0.5) (physics.py:44,69,111,159,330)."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def gaussian_filter(F, sigma, mode="reflect"):
    """scipy.ndimage.gaussian_filter: axis 0 first, then axis 1; per-axis sigma
    and mode allowed (topography.py:164 uses mode=('nearest','wrap'))."""
    sig = (sigma, sigma) if np.isscalar(sigma) else tuple(sigma)
    md = (mode, mode) if isinstance(mode, str) else tuple(mode)
    out = np.asarray(F, dtype=np.float64)
    for ax in (0, 1):
        if sig[ax] > 1e-15:
            out = correlate1d_sym(out, gaussian_kernel1d(sig[ax]), ax, md[ax])
    return out


def conv3_axis(F, axis, mode):
    """scipy.ndimage.convolve(F, [[.25,.5,.25]] or its transpose, mode=...)
    (dynamics.py:229-230, ocean.py:162-163).  N-D correlate accumulates in
    footprint order: ((0 + x[-1]*.25) + x[0]*.5) + x[+1]*.25."""
    n = F.shape[axis]
    base = np.arange(n)
    lo = _take(F, _edge_index(base - 1, n, mode), axis)
    hi = _take(F, _edge_index(base + 1, n, mode), axis)
    return (lo * 0.25 + F * 0.5) + hi * 0.25


def shapiro(F, n=2):
    """SpectralModel._shapiro_filter (dynamics.py:215-231) /
    WindDrivenSlabOcean._shapiro_filter (ocean.py:154-164)."""
    try:
        n = max(1, int(n))
    except Exception:
        n = 2
    out = np.nan_to_num(F, copy=True)
    for _ in range(n):
        out = conv3_axis(out, 1, "wrap")
        out = conv3_axis(out, 0, "nearest")
    return out


def median_positive(x, default):
    """median of the strictly positive entries, `default` when there are none
    (dynamics.py:344-348, physics.py:298-301, run_simulation.py:1866-1874)."""
    pos = x[x > 0]
    if pos.size == 0:
        return float(default)
    return float(np.median(pos))
