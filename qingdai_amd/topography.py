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


# ------------------------------------------------------------------------------------- topography files
def export_topography_to_netcdf(path, grid, land_mask, base_albedo, friction, elevation=None, title="Qingdai Topography"):
    """data/topography.nc (run_simulation.py:126-159; pygcm/topography.py:349-426): lat, lon (f4),
    land_mask (u1), base_albedo, friction (f4), optional elevation (f4)."""
    from . import ncio
    v = {"lat": ("f4", ("lat",), np.asarray(grid.lat, np.float32)), "lon": ("f4", ("lon",), np.asarray(grid.lon, np.float32)),
         "land_mask": ("u1", ("lat", "lon"), np.asarray(land_mask, np.uint8)),
         "base_albedo": ("f4", ("lat", "lon"), np.asarray(base_albedo, np.float32)),
         "friction": ("f4", ("lat", "lon"), np.asarray(friction, np.float32))}
    if elevation is not None:
        v["elevation"] = ("f4", ("lat", "lon"), np.asarray(elevation, np.float32))
    ncio.write_nc(path, {"lat": grid.n_lat, "lon": grid.n_lon}, v, {"title": title, "source": "qingdai_amd", "format": "v1"})


def _regrid(src_lat, src_lon, field, grid, nearest):
    """Bilinear (nearest for the mask) onto the model grid, cyclic in longitude, latitudes clamped to the
    source range; non-finite bilinear results fall back to nearest (pygcm/topography.py:485-520)."""
    from scipy.interpolate import RegularGridInterpolator
    lon3 = np.concatenate([src_lon - 360.0, src_lon, src_lon + 360.0])
    f3 = np.concatenate([field, field, field], axis=1).astype(float)
    pts = np.stack([np.clip(grid.lat_mesh.ravel(), src_lat.min(), src_lat.max()), grid.lon_mesh.ravel()], axis=-1)

    def run(method):
        return RegularGridInterpolator((src_lat, lon3), f3, bounds_error=False, fill_value=None, method=method)(pts).reshape(
            grid.lat_mesh.shape)
    if nearest:
        return np.where(run("nearest") >= 0.5, 1, 0).astype(np.uint8)
    vals = run("linear")
    bad = ~np.isfinite(vals)
    if bad.any():
        vals = np.where(bad, run("nearest"), vals)
    return vals


def load_topography_from_netcdf(path, grid, regrid="auto", quiet=False):
    """-> (elevation | None, land_mask u8, base_albedo, friction) on `grid` (pygcm/topography.py:428-575).
    Source longitudes are brought to [0, 360) and sorted, descending latitudes flipped, a duplicated 0/360
    seam column dropped; an exact grid match is taken as is, anything else is regridded unless regrid="never"."""
    from . import ncio
    v, _ = ncio.read_nc(path, ["lat", "lon", "elevation", "land_mask", "base_albedo", "friction"])
    lat, lon = np.asarray(v["lat"], float), np.asarray(v["lon"], float)
    if np.nanmin(lon) < 0.0 or np.nanmax(lon) <= 180.0:
        lon = np.mod(lon, 360.0)
    flip = not np.all(np.diff(lat) > 0)
    if flip:
        lat = lat[::-1]
    order = np.argsort(lon)
    lon = lon[order]

    def field(name):
        if name not in v:
            return None
        a = np.asarray(v[name])
        a = a[::-1, :] if flip else a
        return a[:, order]
    F = {k: field(k) for k in ("elevation", "land_mask", "base_albedo", "friction")}
    if lon.size >= 2 and np.isclose(lon[0], 0.0) and np.isclose(lon[-1], 360.0):
        lon = lon[:-1]
        F = {k: (a[:, :-1] if a is not None else None) for k, a in F.items()}
    exact = F["land_mask"].shape == (grid.n_lat, grid.n_lon) and (
        regrid == "never" or (np.allclose(lat, grid.lat, atol=1e-6) and np.allclose(lon, grid.lon, atol=1e-6)))
    if exact:
        elev = None if F["elevation"] is None else F["elevation"].astype(float)
        mask, alb, fric = F["land_mask"].astype(np.uint8), F["base_albedo"].astype(float), F["friction"].astype(float)
    else:
        if regrid == "never":
            raise ValueError(f"topography grid mismatch: source {F['land_mask'].shape} vs target {(grid.n_lat, grid.n_lon)}")
        elev = None if F["elevation"] is None else _regrid(lat, lon, F["elevation"], grid, False)
        mask = _regrid(lat, lon, F["land_mask"], grid, True)
        alb, fric = _regrid(lat, lon, F["base_albedo"], grid, False), _regrid(lat, lon, F["friction"], grid, False)
    if not quiet:
        w = np.cos(np.deg2rad(grid.lat_mesh))
        print(f"[Topo] Loaded: {path}")
        print(f"[Topo] Land fraction (achieved): {float((w * (mask == 1)).sum() / (w.sum() + 1e-15)):.3f}")
    return elev, mask, alb, fric
