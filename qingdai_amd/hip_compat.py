"""
qingdai_amd/hip_compat.py -- the operator seam of pygcm/jax_compat.py (jax_compat.py:66-216; SURVEY.md
8(a) a20, 8(b) seam 2) with the MI355X library behind it: same function names, argument lists and return
conventions, so `dynamics.py:95-101,151-157,182-188` / `ocean.py:105-109,125-129,171-176` can route to it
the way they route to the JAX backend.

    is_enabled()            -> True when QD_USE_HIP=1 (default) and libqingdai_hip.so + a device are usable
    backend()               -> "hip" | "numpy"
    to_numpy(x)             -> writeable ndarray (never a device object)
    laplacian_sphere(F, dlat, dlon, coslat, a)
    hyperdiffuse(F, k4, dt, n_substeps, dlat, dlon, coslat, a)
    advect_semilag(field, u, v, dt, a, dlat, dlon, coslat)

The reference passes the ALREADY FLOORED cos(lat) map (0.2 for the atmosphere's Laplacian, 0.5 for the ocean,
1e-6 for the atmosphere's advection): the floor kind is recognised from its minimum, and `a`, `dlat`, `dlon`
must be the grid's own (they are: every caller passes self.grid / constants).  Anything else raises -- this seam
never silently computes something different, and it has no CPU fallback.
"""
from __future__ import annotations

import os

import numpy as np

_handles = {}


def is_enabled() -> bool:
    if int(os.getenv("QD_USE_HIP", "1")) != 1:
        return False
    try:
        from . import _lib
        return _lib.load().qd_device_count() > 0
    except Exception:
        return False


def backend() -> str:
    return "hip" if is_enabled() else "numpy"


def to_numpy(x):
    a = np.asarray(x)
    return a if a.flags.writeable else a.copy()


def _ops(shape, a):
    from . import SphericalGrid
    key = (tuple(shape), float(a))
    if key not in _handles:
        _handles[key] = SphericalGrid(shape[0], shape[1])._ops()
    return _handles[key]


def _kind(coslat, allowed):
    m = float(np.min(coslat))
    for floor, name in allowed:
        if abs(m - floor) <= 1e-12 * max(1.0, floor) or (floor < 1e-3 and m <= 1e-3):
            return name
    raise ValueError(f"hip_compat: unrecognised cos(lat) floor {m!r}; expected one of {[f for f, _ in allowed]}")


def _check_grid(dev, shape, dlat, dlon, a):
    g = dev.grid
    if not (np.isclose(dlat, g.dlat_rad, rtol=1e-12) and np.isclose(dlon, g.dlon_rad, rtol=1e-12)):
        raise ValueError("hip_compat: dlat/dlon are not those of the (n_lat, n_lon) grid")
    if not np.isclose(a, dev.params.a, rtol=1e-12):
        raise ValueError("hip_compat: planet radius differs from the library's")


def laplacian_sphere(F, dlat, dlon, coslat, a):
    F = np.asarray(F, dtype=np.float64)
    dev = _ops(F.shape, a)
    _check_grid(dev, F.shape, dlat, dlon, a)
    return dev.op_laplacian(F, ocean=_kind(coslat, ((0.2, "atm"), (0.5, "ocn"))) == "ocn")


def hyperdiffuse(F, k4, dt, n_substeps, dlat, dlon, coslat, a):
    F = np.asarray(F, dtype=np.float64)
    if float(dt) <= 0.0 or np.all(np.nan_to_num(np.asarray(k4, dtype=np.float64)) <= 0.0):
        return to_numpy(F)                  # the reference's early-outs hand F back untouched (dynamics.py:190-202)
    dev = _ops(F.shape, a)
    _check_grid(dev, F.shape, dlat, dlon, a)
    ocean = _kind(coslat, ((0.2, "atm"), (0.5, "ocn"))) == "ocn"
    k = k4 if np.isscalar(k4) else np.asarray(k4, dtype=np.float64)
    return dev.op_hyperdiffuse(F, k, float(dt), int(n_substeps), ocean=ocean)


def advect_semilag(field, u, v, dt, a, dlat, dlon, coslat):
    field = np.asarray(field, dtype=np.float64)
    dev = _ops(field.shape, a)
    _check_grid(dev, field.shape, dlat, dlon, a)
    return dev.op_advect(field, u, v, float(dt), ocean=_kind(coslat, ((1e-6, "atm"), (0.5, "ocn"))) == "ocn")
