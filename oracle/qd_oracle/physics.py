"""
oracle/qd_oracle/physics.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

Driver-side per-step diagnostics of the reference (pygcm/physics.py:12-354):
convergence precipitation, hybrid precipitation, cloud-from-precip, cloud
source, orographic factor, dynamic albedo.  `m` is any object exposing
u, v, T_s, cloud_cover, P_cond_flux_last (the AtmosOracle or a namespace).
"""
from __future__ import annotations

import numpy as np

from . import numerics as nx
from .params import PLANET_RADIUS


def diagnose_precipitation(m, grid, D_crit, k_precip, cloud_threshold=0.05, smooth_sigma=1.0):
    """physics.py:12-46"""
    div = grid.divergence(m.u, m.v)
    precip = k_precip * np.maximum(0.0, -(div - D_crit))
    if cloud_threshold is not None and cloud_threshold > 0:
        cc = np.clip(m.cloud_cover, 0.0, 1.0)
        precip = precip * (1.0 / (1.0 + np.exp(-10.0 * (cc - cloud_threshold))))
    if smooth_sigma and smooth_sigma > 0:
        precip = nx.gaussian_filter(precip, smooth_sigma)
    return precip


def cloud_from_precip(precip, C_max=0.95, P_ref=2e-5, smooth_sigma=1.0):
    """physics.py:48-70"""
    C = C_max * np.tanh(precip / (P_ref + 1e-12))
    if smooth_sigma and smooth_sigma > 0:
        C = nx.gaussian_filter(C, smooth_sigma)
    return np.clip(C, 0.0, 1.0)


def parameterize_cloud_cover(m, grid):
    """physics.py:72-114"""
    src = np.zeros_like(m.T_s)
    src = src + 0.5 * np.clip(np.tanh((m.T_s - 285.0) / 12.0), 0.0, 1.0)
    vort = grid.vorticity(m.u, m.v)
    rel = vort / (grid.coriolis_param + 1e-12)
    src = src + 0.4 * np.clip(np.tanh((rel - 0.5) / 2.0), 0.0, 1.0)
    dx = grid.dlon_rad * PLANET_RADIUS * np.maximum(1e-6, np.cos(np.deg2rad(grid.lat_mesh)))
    dy = grid.dlat_rad * PLANET_RADIUS
    gx = (np.roll(m.T_s, -1, axis=1) - np.roll(m.T_s, 1, axis=1)) / (2 * dx)
    gy = (np.roll(m.T_s, -1, axis=0) - np.roll(m.T_s, 1, axis=0)) / (2 * dy)
    tadv = -(m.u * gx + m.v * gy)
    src = src + 0.3 * np.clip(np.tanh(np.abs(tadv) / 2e-5), 0.0, 1.0)
    src = nx.gaussian_filter(src, 1.0)
    return np.clip(src, 0.0, 1.0)


def compute_orographic_factor(grid, elevation, u, v, k_orog=7e-4, cap=2.0, smooth_sigma=1.0):
    """physics.py:116-161"""
    a = PLANET_RADIUS
    cos_lat = np.maximum(np.cos(np.deg2rad(grid.lat_mesh)), 1e-6)
    dx = a * cos_lat * grid.dlon_rad
    dy = a * grid.dlat_rad
    dHdx = (np.roll(elevation, -1, axis=1) - np.roll(elevation, 1, axis=1)) / (2.0 * dx)
    dHdy = (np.roll(elevation, -1, axis=0) - np.roll(elevation, 1, axis=0)) / (2.0 * dy)
    dHdy[0, :] = 0.0
    dHdy[-1, :] = 0.0
    gn = np.sqrt(dHdx ** 2 + dHdy ** 2)
    with np.errstate(all="ignore"):
        nxh = np.where(gn > 1e-12, dHdx / (gn + 1e-12), 0.0)
        nyh = np.where(gn > 1e-12, dHdy / (gn + 1e-12), 0.0)
    factor = np.clip(1.0 + k_orog * np.maximum(0.0, u * nxh + v * nyh), 1.0, cap)
    if smooth_sigma and smooth_sigma > 0:
        factor = nx.gaussian_filter(factor, smooth_sigma)
    return factor


def calculate_dynamic_albedo(cloud_cover, T_s, base_albedo, alpha_ice, alpha_cloud, land_mask=None,
                             t_freeze=271.35, delta_T=5.0, ice_frac=None):
    """physics.py:164-250 (ice_frac given | temperature tanh transition; ice only over ocean)."""
    T = np.asarray(T_s, dtype=float)
    C = np.clip(np.asarray(cloud_cover, dtype=float), 0.0, 1.0)
    base = base_albedo.astype(float) if isinstance(base_albedo, np.ndarray) else np.full_like(T, float(base_albedo))
    if ice_frac is not None:
        fi = np.clip(np.asarray(ice_frac, dtype=float), 0.0, 1.0)
    else:
        fi = 0.5 * (1.0 + np.tanh((t_freeze - T) / max(1e-6, float(delta_T))))
    if land_mask is not None:
        fi = fi * (land_mask == 0)
    elif isinstance(base_albedo, np.ndarray):
        fi = fi * (base < 0.15)
    surf = base * (1.0 - fi) + float(alpha_ice) * fi
    return np.clip(surf * (1.0 - C) + float(alpha_cloud) * C, 0.0, 1.0)


def diagnose_precipitation_hybrid(m, grid, P, orog_factor=None, smooth_sigma=1.0, renorm=True):
    """physics.py:253-354 with D_crit/k_precip/beta_div/fallback knobs from `P`."""
    Pq = np.maximum(0.0, np.asarray(m.P_cond_flux_last, dtype=float))
    div = grid.divergence(m.u, m.v)
    pos = np.maximum(0.0, -(div - float(P.D_crit)))
    if np.any(pos > 0):
        scale = max(float(np.median(pos[pos > 0])), 1e-12)
        F_div = np.clip(pos / scale, 0.0, 5.0)
    else:
        F_div = np.zeros_like(Pq)
    F_orog = 1.0 if orog_factor is None else np.clip(np.asarray(orog_factor, dtype=float), 1.0, 3.0)
    F = (1.0 + float(P.p_betadiv) * F_div) * F_orog
    P_raw = Pq * F
    w = np.maximum(np.cos(np.deg2rad(grid.lat_mesh)), 0.0)
    if renorm:
        num = float(np.sum(Pq * w))
        den = float(np.sum(P_raw * w)) + 1e-20
        s = num / den if den > 0 else 1.0
        Pr = P_raw * s
    else:
        Pr = P_raw
    if smooth_sigma and smooth_sigma > 0:
        Pr = nx.gaussian_filter(Pr, float(smooth_sigma))
    if P.p_hybrid_fallback:
        wsum = float(np.sum(w) + 1e-15)
        Pq_mean = float(np.sum(Pq * w) / wsum)
        if Pq_mean < P.pq_min:
            P_dyn = diagnose_precipitation(m, grid, P.D_crit, P.k_precip, cloud_threshold=None,
                                           smooth_sigma=smooth_sigma)
            Pr = (1.0 - P.p_blend) * Pr + P.p_blend * P_dyn
    return np.clip(Pr, 0.0, None)
