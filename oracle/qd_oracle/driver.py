"""
oracle/qd_oracle/driver.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

The driver-side per-step physics of scripts/run_simulation.py:1766-1934 and 2063-2146,
restated as one function over the oracle's own operators (ecology, phytoplankton, orography
and the snow-albedo blend off / absent: C_snow = 0):
   hybrid precipitation -> cloud-from-precip -> cloud source -> cloud blend + floor ->
   cloud advection (cos floor 0.5) + sigma=0.2 wrap blur -> dynamic albedo.
"""
from __future__ import annotations

import numpy as np

from . import numerics as nx
from . import physics as ph
from .atmos import advect_semilag
from .params import is_set


def driver_physics_step(m, grid, P, base_albedo, land_mask, dt):
    """Mutates m.cloud_cover; returns (precip, albedo).  `m` is an AtmosOracle."""
    precip = ph.diagnose_precipitation_hybrid(m, grid, P, orog_factor=None, smooth_sigma=1.0, renorm=True)
    # run_simulation.py:1866-1881
    if np.any(precip > 0):
        if is_set(P.pref) and P.pref != 0.0:
            P_ref = float(P.pref)
        else:
            P_ref = nx.median_positive(precip, 1e-6)
    else:
        P_ref = 1e-6
    C_from_P = ph.cloud_from_precip(precip, C_max=float(P.cmax), P_ref=P_ref, smooth_sigma=1.0)
    src = ph.parameterize_cloud_cover(m, grid)
    # run_simulation.py:1890-1913
    tendency = src * (dt / (6 * 3600))
    W_MEM, W_P, W_SRC = float(P.w_mem), float(P.w_p), float(P.w_src)
    W_sum = W_MEM + W_P + W_SRC
    if W_sum <= 0:
        W_MEM, W_P, W_SRC, W_sum = 0.5, 0.4, 0.1, 1.0
    W_MEM /= W_sum
    W_P /= W_sum
    W_SRC /= W_sum
    cc = (W_MEM * m.cloud_cover + W_P * C_from_P + W_SRC * np.clip(m.cloud_cover + tendency, 0.0, 1.0))
    if P.cloud_from_p_floor > 0.0:
        cc = np.maximum(cc, np.clip(P.cloud_from_p_floor * C_from_P, 0.0, 1.0))
    cc = np.clip(cc, 0.0, 1.0)
    # run_simulation.py:1916-1934
    if P.cloud_advect:
        cos05 = np.maximum(np.cos(np.deg2rad(grid.lat_mesh)), 0.5)
        adv = advect_semilag(cc, m.u, m.v, dt, 6.371e6, grid.dlat_rad, grid.dlon_rad, cos05)
        if P.cloud_smooth_sigma > 0.0:
            adv = nx.gaussian_filter(adv, P.cloud_smooth_sigma, "wrap")
        al = float(P.cloud_adv_alpha)
        cc = np.clip((1.0 - al) * cc + al * adv, 0.0, 1.0)
    m.cloud_cover = cc
    # run_simulation.py:2063-2146 (no ecology / phyto / snow)
    ice_frac = 1.0 - np.exp(-np.maximum(m.h_ice, 0.0) / max(1e-6, P.hice_ref))
    cloud_for_rad = m.cloud_eff_last if getattr(m, "cloud_eff_last", None) is not None else m.cloud_cover
    base_in = base_albedo.copy() if P.use_topo_albedo else np.full_like(m.T_s, float(P.alpha_water))
    albedo = ph.calculate_dynamic_albedo(cloud_for_rad, m.T_s, base_in, P.alpha_ice, P.alpha_cloud,
                                         land_mask=land_mask, ice_frac=ice_frac)
    return precip, albedo
