"""
oracle/qd_oracle/column.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

Per-cell column physics of the reference restated in NumPy:
humidity (pygcm/humidity.py:85-183) and the explicit surface/atmosphere energy
budget (pygcm/energy.py:77-491).  Pure functions of arrays + the flat parameter
namespace of params.py; operation order follows the reference expression by
expression so results agree to the last bit wherever libm agrees.
"""
from __future__ import annotations

import numpy as np

from .params import SIGMA_SB

EPSILON = 0.622  # humidity.py:34


def q_sat(T, p0):
    """humidity.py:85-101 (Tetens, T_c clipped to [-80, 60], q<=0.5)."""
    T_c = np.clip(np.asarray(T, dtype=float) - 273.15, -80.0, 60.0)
    e_s = 610.94 * np.exp(17.625 * T_c / (T_c + 243.04))
    denom = np.maximum(p0 - (1.0 - EPSILON) * e_s, 1.0)
    return np.clip(EPSILON * e_s / denom, 0.0, 0.5)


def q_init(Ts, RH0, p0):
    """humidity.py:104-113"""
    return float(np.clip(RH0, 0.0, 1.0)) * q_sat(Ts, p0)


def surface_evaporation_factor(land_mask, h_ice, P):
    """humidity.py:116-142 (ice threshold 1e-6)."""
    land = (land_mask == 1)
    ocean = ~land
    fac = np.zeros(land_mask.shape, dtype=float)
    ice = (h_ice > 1e-6) & ocean
    fac[ice] = float(P.ice_evap_scale)
    fac[ocean & ~ice] = float(P.ocean_evap_scale)
    fac[land] = float(P.land_evap_scale)
    return fac


def evaporation_flux(Ts, q, u, v, fac, P):
    """humidity.py:145-159"""
    V = np.sqrt(u ** 2 + v ** 2)
    deficit = np.maximum(0.0, q_sat(Ts, P.p0) - q)
    return np.nan_to_num(P.rho_a * P.C_E * V * deficit * fac)


def condensation(q, T_a, dt, P):
    """humidity.py:162-183 -> (P_cond_flux, q_next)"""
    excess = np.maximum(0.0, q - q_sat(T_a, P.p0))
    M_col = max(1e-6, float(P.rho_a * P.h_mbl))
    P_cond = (excess / max(1e-6, float(P.tau_cond))) * M_col
    q_next = q - (P_cond / M_col) * dt
    q_next = np.clip(np.nan_to_num(q_next), 0.0, 0.5)
    return np.nan_to_num(P_cond), q_next


def shortwave(I, albedo, cloud, P):
    """energy.py:77-98 -> (SW_atm, SW_sfc, R)"""
    alpha = np.clip(albedo, 0.0, 1.0)
    Ic = np.maximum(0.0, I)
    R = Ic * alpha
    A_sw = np.clip(P.sw_a0 + P.sw_kc * np.clip(cloud, 0.0, 1.0), 0.0, 0.95)
    SW_atm = Ic * A_sw
    SW_sfc = np.maximum(0.0, Ic - R - SW_atm)
    return SW_atm, SW_sfc, R


def longwave_v1(Ts, Ta, cloud, P):
    """energy.py:101-137 -> (LW_atm, LW_sfc, OLR, DLR, eps)"""
    s = SIGMA_SB
    Ts4 = np.maximum(0.0, Ts) ** 4
    Ta4 = np.maximum(0.0, Ta) ** 4
    eps = np.clip(P.lw_eps0 + P.lw_kc * np.clip(cloud, 0.0, 1.0), 0.0, 1.0)
    OLR = eps * s * Ta4 + (1.0 - eps) * s * Ts4
    DLR = eps * s * Ta4
    LW_sfc = DLR - s * Ts4
    LW_atm = eps * (s * Ts4 - 2.0 * s * Ta4)
    if P.gh_lock:
        g = P.gh_factor_lw
        OLR = (1.0 - g) * s * Ts4
        DLR = g * s * Ts4
        LW_sfc = DLR - s * Ts4
    return LW_atm, LW_sfc, OLR, DLR, eps


def surface_emissivity_map(land_mask, ice_frac, P):
    """energy.py:141-158"""
    land = (land_mask == 1)
    ocean = ~land
    eps = np.full(ice_frac.shape, P.eps_land, dtype=float)
    fi = np.clip(ice_frac[ocean], 0.0, 1.0)
    eps[ocean] = (1.0 - fi) * P.eps_ocean + fi * P.eps_ice
    return np.nan_to_num(eps)


def longwave_v2(Ts, Ta, cloud_eff, eps_sfc, P):
    """energy.py:161-234 -> (LW_atm, LW_sfc, OLR, DLR, eps_eff)"""
    s = SIGMA_SB
    Ts = np.maximum(0.0, Ts)
    Ta = np.maximum(0.0, Ta)
    Ts4 = Ts ** 4
    Ta4 = Ta ** 4
    eps_clear = np.clip(float(P.lw_eps0), 0.0, 1.0)
    ce = np.clip(cloud_eff, 0.0, 1.0)
    tau_cloud = P.lw_tau0 * ce
    eps_cloud = np.clip(1.0 - np.exp(-P.lw_ktau * tau_cloud), 0.0, 1.0)
    eps_eff = 1.0 - (1.0 - eps_clear) * (1.0 - eps_cloud)
    if np.isscalar(eps_sfc):
        es = np.full_like(Ts, float(eps_sfc))
    else:
        es = np.clip(np.nan_to_num(eps_sfc), 0.0, 1.0)
    OLR = eps_eff * s * Ta4 + (1.0 - eps_eff) * s * es * Ts4
    DLR = eps_eff * s * Ta4
    LW_sfc = DLR - s * es * Ts4
    LW_atm = eps_eff * (s * es * Ts4 - 2.0 * s * Ta4)
    if P.gh_lock:
        g = P.gh_factor_lw
        Ts4_raw = np.maximum(0.0, Ts) ** 4
        OLR = (1.0 - g) * s * Ts4_raw
        DLR = g * s * Ts4_raw
        LW_sfc = DLR - s * es * Ts4
    return LW_atm, LW_sfc, OLR, DLR, eps_eff


def sensible_heat(Ts, Ta, u, v, P):
    """energy.py:423-442 (SH only; the Bowen-ratio LH is unused by the path)."""
    V = np.sqrt(u ** 2 + v ** 2)
    return P.rho_a * P.cp_a * P.ch * V * (Ts - Ta)


def integrate_surface_energy(Ts, SW_sfc, LW_sfc, SH, LH, dt, P):
    """energy.py:237-260 (scalar heat capacity QD_CS)."""
    net = SW_sfc - LW_sfc - SH - LH
    Ts_next = Ts + (net / max(1e-12, P.c_sfc)) * dt
    return np.nan_to_num(np.maximum(P.t_floor, Ts_next))


def integrate_surface_energy_map(Ts, SW_sfc, LW_sfc, SH, LH, dt, C_s_map, P):
    """energy.py:263-288"""
    net = SW_sfc - LW_sfc - SH - LH
    Cs = np.where(np.isfinite(C_s_map) & (C_s_map > 1e3), C_s_map, 1e3)
    Ts_next = Ts + (net / Cs) * dt
    return np.nan_to_num(np.maximum(P.t_floor, Ts_next))


def integrate_surface_energy_with_seaice(Ts, SW_sfc, LW_sfc, SH, LH, dt, land_mask, h_ice, P):
    """energy.py:291-420 -> (Ts_next, h_ice_next).  Melt first, then freeze,
    residual heats with Cs_eff, polar-row freeze fix, ice-surface cap, floor."""
    Q = SW_sfc - LW_sfc - SH - LH
    land = (land_mask == 1)
    ocean = ~land
    Ts_n = Ts.astype(float).copy()
    hi = h_ice.astype(float).copy()
    rL = P.rho_i * P.L_f

    melt = (hi > 0.0) & ocean & (Q > 0.0)
    if np.any(melt):
        dh_melt = (Q[melt] * dt) / rL
        dh_cap = np.minimum(dh_melt, hi[melt])
        hi[melt] -= dh_cap
        Q[melt] = Q[melt] - (dh_cap * P.rho_i * P.L_f) / dt

    frz = ocean & (Q < 0.0) & (Ts_n <= (P.t_freeze + 0.5))
    if np.any(frz):
        hi[frz] += (-Q[frz] * dt) / rL
        Q[frz] = 0.0
        Ts_n[frz] = np.minimum(Ts_n[frz], P.t_freeze)

    Cs_eff = np.where(land, P.Cs_land, np.where(hi > 0.0, P.Cs_ice, P.Cs_ocean))
    Cs_eff = np.where(np.isfinite(Cs_eff) & (Cs_eff > 1e3), Cs_eff, 1e3)
    Ts_n = Ts_n + (Q / Cs_eff) * dt

    for enabled, j in ((P.polar_freeze_fix_s, 0), (P.polar_freeze_fix_n, -1)):
        if enabled:
            m = ocean[j, :] & (Q[j, :] < 0.0) & (Ts_n[j, :] > P.t_freeze)
            if np.any(m):
                Ts_n[j, m] = P.t_freeze

    Ts_n = np.where((hi > 0.0) & ocean, np.minimum(Ts_n, P.t_freeze), Ts_n)
    Ts_n = np.maximum(P.t_floor, Ts_n)
    return np.nan_to_num(Ts_n), np.nan_to_num(hi)


def integrate_atmos_energy_height(h, SW_atm, LW_atm, SH, LH_release, dt, rho_air, H_atm, g, weight):
    """energy.py:452-491"""
    F_atm = SW_atm + LW_atm + SH + LH_release
    denom = max(1e-6, float(rho_air)) * max(1.0, float(H_atm)) * float(g)
    return np.nan_to_num(h + float(weight) * (F_atm / denom) * dt)


def energy_diagnostics(lat_mesh, I, R, OLR, SW_sfc, LW_sfc, SH, LH):
    """energy.py:494-538 (cos-weighted global means)."""
    TOA = I - R - OLR
    SFC = SW_sfc - LW_sfc - SH - LH
    ATM = TOA - SFC
    w = np.maximum(np.cos(np.deg2rad(lat_mesh)), 0.0)
    ws = np.sum(w)

    def wm(x):
        return float(np.sum(x * w) / (ws + 1e-15))
    return {"TOA_net": wm(TOA), "SFC_net": wm(SFC), "ATM_net": wm(ATM), "I_mean": wm(I),
            "R_mean": wm(R), "OLR_mean": wm(OLR), "SW_sfc_mean": wm(SW_sfc),
            "LW_sfc_mean": wm(LW_sfc), "SH_mean": wm(SH), "LH_mean": wm(LH)}


def autotune_greenhouse(eps0, kc, diag, rate_eps=5e-5, rate_kc=2e-5, bounds_eps=(0.30, 0.98), bounds_kc=(0.0, 0.80)):
    """energy.py:544-579: proportional nudge of (lw_eps0, lw_kc) by TOA_net; returns the new pair."""
    err = float(diag.get("TOA_net", 0.0))
    return (float(np.clip(eps0 - rate_eps * err, bounds_eps[0], bounds_eps[1])),
            float(np.clip(kc - rate_kc * err, bounds_kc[0], bounds_kc[1])))
