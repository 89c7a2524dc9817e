"""
TEST INFRASTRUCTURE ONLY.  Transport of the phytoplankton tracers by the ocean currents:
pygcm/ecology/phyto.py:452-547 (`PhytoManager.advect_diffuse` with its own `_advect_scalar` /
`_laplacian_sphere`, both on the ocean cos floor 0.5).
"""
from __future__ import annotations

import numpy as np

from .atmos import advect_semilag, laplacian_sphere
from .grid import PLANET_RADIUS


def advect_diffuse(C_s, uo, vo, dt_seconds, grid, land_mask, K_h=5.0e3, adv_alpha=0.7):
    """C_s: [S, n_lat, n_lon] -> new array.  Semi-Lagrangian blend, explicit lateral diffusion, clip >= 0,
    land zero, polar-ring mean over the ocean cells of the first and last row."""
    C_s = np.array(C_s, dtype=float, copy=True)
    if dt_seconds <= 0.0:
        return C_s
    coslat = np.maximum(np.cos(np.deg2rad(grid.lat_mesh)), 0.5)
    ocean = (land_mask == 0)
    for s in range(C_s.shape[0]):
        C = C_s[s]
        C_adv = advect_semilag(C, uo, vo, float(dt_seconds), PLANET_RADIUS, grid.dlat_rad, grid.dlon_rad, coslat)
        C_new = (1.0 - adv_alpha) * C + adv_alpha * C_adv
        if K_h > 0.0:
            C_new = np.nan_to_num(C_new)
            C_new += float(dt_seconds) * K_h * laplacian_sphere(C_new, grid.dlat_rad, grid.dlon_rad, coslat, PLANET_RADIUS)
        C_new = np.clip(C_new, 0.0, np.inf)
        C_new[~ocean] = 0.0
        C_s[s] = C_new
    for j in (0, -1):
        row = ocean[j, :]
        if np.any(row):
            for s in range(C_s.shape[0]):
                C_s[s, j, row] = float(np.mean(C_s[s, j, :][row]))
    return C_s
