"""
qingdai_amd/topography.py -- host-side, init-time only (NumPy).

Procedural seed-42 land/sea mask and base albedo / friction maps with the same
recipe as the reference (pygcm/topography.py:92-346): three generalized-Gaussian
continents blended with very-low-frequency noise, five fBm octaves, a cos(lat)
weighted quantile sea level hitting the target land fraction.  It exists so the
benchmark configurations of BASELINE.json ("default seed=42 topography") can be
built at any resolution without the reference present.  Not on the per-step path.
"""
from __future__ import annotations

import numpy as np


def _kernel(sigma, truncate=4.0):
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1)
    k = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return k / k.sum()


def _filter_axis(F, sigma, axis, mode):
    """Separable Gaussian pass; mode 'nearest' (clamp) or 'wrap' (period n).
    Tap order mirrors scipy's symmetric correlate1d so maps agree bit for bit."""
    w = _kernel(sigma)
    r = (len(w) - 1) // 2
    n = F.shape[axis]
    base = np.arange(n)

    def ext(i):
        return np.mod(i, n) if mode == "wrap" else np.clip(i, 0, n - 1)
    out = np.take(F, base, axis=axis) * w[r]
    for j in range(-r, 0):
        out = out + (np.take(F, ext(base + j), axis=axis) + np.take(F, ext(base - j), axis=axis)) * w[r + j]
    return out


def _smooth(F, sig_lat, sig_lon):
    return _filter_axis(_filter_axis(F, sig_lat, 0, "nearest"), sig_lon, 1, "wrap")


def _norm(x):
    return (x - x.mean()) / (x.std() + 1e-8)


def generate_elevation_map(grid, seed=42):
    lat_mesh, lon_mesh = grid.lat_mesh, grid.lon_mesh
    n_lat, n_lon = lat_mesh.shape
    # L1 continents
    rng = np.random.default_rng(int(seed))
    c_lat = np.rad2deg(np.arcsin(rng.uniform(-1.0, 1.0, size=3)))
    c_lon = rng.uniform(0.0, 360.0, size=3)
    c_amp = rng.uniform(0.8, 1.2, size=3)
    lat = np.deg2rad(lat_mesh)
    lon = np.deg2rad(lon_mesh)
    H1 = np.zeros_like(lat_mesh, dtype=float)
    sig = np.deg2rad(30.0)
    for la0, lo0, A in zip(c_lat, c_lon, c_amp):
        la0r, lo0r = np.deg2rad(la0), np.deg2rad(lo0)
        cosd = np.clip(np.sin(lat) * np.sin(la0r) + np.cos(lat) * np.cos(la0r) * np.cos(lon - lo0r), -1.0, 1.0)
        H1 += A * np.exp(-(np.arccos(cosd) / sig) ** 2.0)
    H1 = _norm(H1)
    noise = rng.standard_normal(size=(n_lat, n_lon))
    vlf = _norm(_smooth(noise, float(max(4, n_lat // 12)), float(max(8, n_lon // 12))))
    H1 = _norm((1 - 0.35) * H1 + 0.35 * vlf)
    # L3 fBm
    rng3 = np.random.default_rng(int(seed) + 1)
    fbm = np.zeros((n_lat, n_lon))
    amp = 1.0
    s_lat = float(max(1, n_lat // 20))
    s_lon = float(max(1, n_lon // 20))
    for _ in range(5):
        layer = _norm(_smooth(rng3.standard_normal(size=(n_lat, n_lon)), s_lat, s_lon))
        fbm += amp * layer
        amp *= 2 ** (-0.8)
        s_lat = max(0.5, s_lat / 2.0)
        s_lon = max(0.5, s_lon / 2.0)
    fbm = _norm(fbm)
    elev = _norm(1.0 * H1 + 0.6 * fbm) * 4500.0
    return _smooth(elev, 0.5, 0.5)


def create_land_sea_mask(grid, target_land_frac=0.29, seed=42, return_elevation=False):
    elev = generate_elevation_map(grid, seed=seed)
    w = np.maximum(np.cos(np.deg2rad(grid.lat_mesh)), 0.0).ravel()
    v = elev.ravel()
    order = np.argsort(v)
    vs, ws = v[order], w[order]
    cw = np.cumsum(ws)
    cw /= cw[-1]
    idx = int(np.clip(np.searchsorted(cw, 1.0 - float(target_land_frac), side="left"), 0, v.size - 1))
    mask = (elev >= float(vs[idx])).astype(np.uint8)
    return (mask, elev) if return_elevation else mask


def generate_base_properties(mask):
    """Ice-free base albedo / Rayleigh friction maps (reference defaults with
    elevation=None, grid=None): ocean 0.08 / 1e-6, land 0.28 / 1e-5."""
    mask = mask.astype(np.uint8)
    albedo = np.clip(np.where(mask == 1, 0.28, 0.08), 0.05, 0.85)
    friction = np.clip(np.where(mask == 1, 1.0e-5, 1.0e-6), 5e-7, 3e-5)
    return albedo, friction
