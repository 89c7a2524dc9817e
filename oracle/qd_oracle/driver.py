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


def orog_factor_of(m, grid, P, elevation):
    """run_simulation.py:1769-1775: only with QD_OROG=1 and an elevation map."""
    if int(getattr(P, "orog_enable", 0)) == 1 and elevation is not None:
        return ph.compute_orographic_factor(grid, elevation, m.u, m.v, k_orog=float(P.orog_k))
    return None


def driver_physics_step(m, grid, P, base_albedo, land_mask, dt, elevation=None):
    """Mutates m.cloud_cover; returns (precip, albedo).  `m` is an AtmosOracle."""
    precip = ph.diagnose_precipitation_hybrid(m, grid, P, orog_factor=orog_factor_of(m, grid, P, elevation),
                                              smooth_sigma=1.0, renorm=True)
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


# =============================================================================================
# hydrology (pygcm/hydrology.py) and the whole driver step (scripts/run_simulation.py:1760-2340)
# =============================================================================================
def partition_precip_phase_smooth(P_flux, T_hat_a, T_thresh, dT_half):
    """hydrology.py:100-122"""
    f_snow = 1.0 / (1.0 + np.exp((T_hat_a - float(T_thresh)) / max(1e-6, float(dT_half))))
    f_snow = np.clip(f_snow, 0.0, 1.0)
    return np.nan_to_num((1.0 - f_snow) * P_flux), np.nan_to_num(f_snow * P_flux), f_snow


def snowpack_step(S_snow, P_snow_land, T_hat_a, P, dt):
    """hydrology.py:124-177 -> (S_next, melt_flux, C_snow, alpha_snow)"""
    S = np.asarray(S_snow, dtype=float).copy()
    if P.snow_melt_mode == 0:
        melt_flux = (float(P.snow_ddf_mm_per_k_day) / 86400.0) * np.maximum(T_hat_a - float(P.snow_melt_tref_K), 0.0)
    else:
        melt_flux = np.where(T_hat_a >= float(P.snow_thresh_K), float(P.snow_melt_rate_mm_day) / 86400.0, 0.0)
    actual = np.minimum(np.maximum(S, 0.0), melt_flux * dt)
    S_next = S + P_snow_land * dt - actual
    if is_set(P.swe_max_mm) and P.swe_max_mm > 0:
        S_next = np.minimum(S_next, float(P.swe_max_mm))
    S_next = np.maximum(0.0, S_next)
    melt_out = np.where(dt > 0, actual / dt, 0.0)
    C_snow = np.clip(1.0 - np.exp(-np.maximum(S_next, 0.0) / max(1e-6, float(P.swe_ref_mm))), 0.0, 1.0)
    return np.nan_to_num(S_next), np.nan_to_num(melt_out), C_snow, np.full_like(S_next, float(P.snow_albedo_fresh))


def update_land_bucket(W_land, P_in, E_land, P, dt):
    """hydrology.py:219-260 -> (W_next, R_flux)"""
    W = np.asarray(W_land, dtype=float).copy()
    tau_s = max(1.0, float(P.runoff_tau_days) * 86400.0)
    R_base = W / tau_s
    W_next = np.maximum(0.0, W + (P_in - E_land - R_base) * dt)
    if is_set(P.wland_cap_mm) and P.wland_cap_mm > 0:
        overflow = np.maximum(0.0, W_next - float(P.wland_cap_mm))
        W_next = W_next - overflow
        R_fast = np.where(dt > 0, overflow / dt, 0.0)
    else:
        R_fast = 0.0
    return np.nan_to_num(W_next), np.nan_to_num(R_base + R_fast)


class DriverOracle:
    """One iteration of the reference driver loop (ecology / phytoplankton / routing / plots off):
    run_simulation.py:1766-1934 (precipitation, clouds), 1942-2019 (insolation, P019 lapse + snow),
    2063-2146 (albedo), 2191-2194 (Teq, time_step WITHOUT albedo), 2197-2253 (ocean coupling),
    2290-2339 (snow commit + land bucket)."""

    def __init__(self, grid, atm, ocean, forcing, land_mask, base_albedo, P, elevation=None):
        self.grid, self.atm, self.ocean, self.forcing = grid, atm, ocean, forcing
        self.land_mask, self.base_albedo, self.P = land_mask, base_albedo, P
        self.elevation = elevation
        shp = grid.lat_mesh.shape
        self.W_land = np.zeros(shp)
        self.S_snow = np.zeros(shp)
        self.C_snow = np.zeros(shp)
        self.precip = np.zeros(shp)
        self.albedo = np.zeros(shp)
        self.R_flux = np.zeros(shp)
        # optional ecology (qd_oracle.ecology): EcoCoupling blended into the base albedo, IndividualSubstep + its bands
        self.eco = None
        self.indiv = None
        self.indiv_bands = None
        self.indiv_day = None
        self.soil_cap = 50.0

    def step(self, t, dt, pass_albedo=False, commit=True):
        """pass_albedo: call time_step(Teq, dt, albedo=albedo) -- the call shape of scripts/benchmark_jax.py:96,132 (BASELINE
        configs[2]; the reference driver itself never passes it).  commit=False leaves S_snow / W_land untouched (no hydrology block)."""
        from . import column as col
        P, m, g = self.P, self.atm, self.grid
        land = (self.land_mask == 1)
        # --- precipitation + clouds (everything of driver_physics_step up to the cloud tracer)
        precip = ph.diagnose_precipitation_hybrid(m, g, P, orog_factor=orog_factor_of(m, g, P, self.elevation),
                                                  smooth_sigma=1.0, renorm=True)
        if np.any(precip > 0):
            P_ref = float(P.pref) if (is_set(P.pref) and P.pref != 0.0) else nx.median_positive(precip, 1e-6)
        else:
            P_ref = 1e-6
        C_from_P = ph.cloud_from_precip(precip, C_max=float(P.cmax), P_ref=P_ref, smooth_sigma=1.0)
        src = ph.parameterize_cloud_cover(m, g)
        W_MEM, W_P, W_SRC = float(P.w_mem), float(P.w_p), float(P.w_src)
        W_sum = W_MEM + W_P + W_SRC
        if W_sum <= 0:
            W_MEM, W_P, W_SRC, W_sum = 0.5, 0.4, 0.1, 1.0
        W_MEM /= W_sum
        W_P /= W_sum
        W_SRC /= W_sum
        cc = (W_MEM * m.cloud_cover + W_P * C_from_P + W_SRC * np.clip(m.cloud_cover + src * (dt / (6 * 3600)), 0.0, 1.0))
        if P.cloud_from_p_floor > 0.0:
            cc = np.maximum(cc, np.clip(P.cloud_from_p_floor * C_from_P, 0.0, 1.0))
        cc = np.clip(cc, 0.0, 1.0)
        if P.cloud_advect:
            cos05 = np.maximum(np.cos(np.deg2rad(g.lat_mesh)), 0.5)
            adv = advect_semilag(cc, m.u, m.v, dt, 6.371e6, g.dlat_rad, g.dlon_rad, cos05)
            if P.cloud_smooth_sigma > 0.0:
                adv = nx.gaussian_filter(adv, P.cloud_smooth_sigma, "wrap")
            cc = np.clip((1.0 - P.cloud_adv_alpha) * cc + P.cloud_adv_alpha * adv, 0.0, 1.0)
        m.cloud_cover = cc
        # --- insolation
        a_, b_ = self.forcing.insolation_components(t)
        m.isr_A, m.isr_B, m.isr = a_, b_, a_ + b_
        # --- P019 lapse + phase split + provisional snowpack (run_simulation.py:1946-2019)
        T_a_proxy = 288.0 + (9.81 / 1004.0) * m.h
        H_bed = self.elevation if self.elevation is not None else np.zeros_like(m.T_s)
        h_snow_geom = np.where(land, np.maximum(self.S_snow, 0.0) / max(P.rho_snow, 1e-6), 0.0)
        polar = (np.abs(g.lat_mesh) >= P.polar_lat_thresh)
        h_ice_eff = np.where(polar, np.minimum(h_snow_geom, P.polar_ice_thick_max_m), h_snow_geom)
        H_eff = np.minimum(H_bed + h_ice_eff, P.land_elev_max_m)
        T_hat_a = T_a_proxy - P.lapse_k_kpm * (H_eff / 1000.0) if P.lapse_enable else T_a_proxy
        P_rain, P_snow, _ = partition_precip_phase_smooth(precip, T_hat_a, P.snow_thresh_K, P.snow_t_band_K)
        if P.swe_enable:
            S_next, melt_land, C_snow, alpha_snow = snowpack_step(self.S_snow, P_snow * land, T_hat_a, P, dt)
            glacier = land & ((C_snow >= P.glacier_frac) | (S_next >= P.glacier_swe_mm))
            P_rain_gl = (P_rain * land) * glacier
            if np.any(P_rain_gl):
                S_next = S_next + P_rain_gl * dt
        else:
            C_snow = np.zeros_like(m.T_s)
            glacier = land & (C_snow >= P.glacier_frac)
            alpha_snow = np.full_like(m.T_s, float(P.snow_albedo_fresh))
            S_next = self.S_snow.copy()
            melt_land = np.zeros_like(m.T_s)
        # --- albedo (run_simulation.py:2063-2146)
        ice_frac = 1.0 - np.exp(-np.maximum(m.h_ice, 0.0) / max(1e-6, P.hice_ref))
        cloud_for_rad = m.cloud_eff_last if getattr(m, "cloud_eff_last", None) is not None else m.cloud_cover
        base_in = self.base_albedo.copy() if P.use_topo_albedo else np.full_like(m.T_s, float(P.alpha_water))
        if self.indiv is not None:                             # run_simulation.py:2021-2046
            soil_idx = np.clip(self.W_land / max(1e-6, self.soil_cap), 0.0, 1.0)
            self.indiv.try_substep(m.isr_A, m.isr_B, self.indiv_bands, soil_idx, dt, self.indiv_day)
        if self.eco is not None:                               # run_simulation.py:2075-2128
            self.eco.apply(base_in, land, glacier, m.isr, dt)
        if P.swe_enable:
            base_in[land] = np.clip((1.0 - C_snow[land]) * base_in[land] + C_snow[land] * alpha_snow[land], 0.0, 1.0)
        albedo = ph.calculate_dynamic_albedo(cloud_for_rad, m.T_s, base_in, P.alpha_ice, P.alpha_cloud,
                                             land_mask=self.land_mask, ice_frac=ice_frac)
        # --- Teq + dynamics (the driver does NOT pass albedo: run_simulation.py:2194)
        Teq = self.forcing.equilibrium_temp(t, albedo)
        if pass_albedo:
            m.time_step(Teq, dt, albedo=albedo)
        else:
            m.time_step(Teq, dt)
        # --- ocean coupling (run_simulation.py:2197-2253)
        if self.ocean is not None:
            ice_mask = m.h_ice > 0.0
            cloud_eff = m.cloud_eff_last if getattr(m, "cloud_eff_last", None) is not None else m.cloud_cover
            _, SW_sfc, _ = col.shortwave(m.isr, albedo, cloud_eff, P)
            T_a = 288.0 + (9.81 / 1004.0) * m.h
            ice_frac2 = 1.0 - np.exp(-np.maximum(m.h_ice, 0.0) / max(1e-6, P.hice_ref))
            if P.lw_v2:
                _, LW_sfc, _, _, _ = col.longwave_v2(m.T_s, T_a, cloud_eff, col.surface_emissivity_map(self.land_mask, ice_frac2, P), P)
            else:
                _, LW_sfc, _, _, _ = col.longwave_v1(m.T_s, T_a, cloud_eff, P)
            SH = col.sensible_heat(m.T_s, T_a, m.u, m.v, P)
            Q_net = SW_sfc - LW_sfc - SH - m.LH_last
            self.ocean.step(dt, m.u, m.v, Q_net=Q_net, ice_mask=ice_mask)
            m.T_s = np.where((self.land_mask == 0) & (~ice_mask), self.ocean.Ts, m.T_s)
        self.C_snow, self.precip, self.albedo = C_snow, precip, albedo
        self.glacier = glacier
        if not commit:
            return
        # --- hydrology commit (run_simulation.py:2290-2339)
        E_land = m.E_flux_last * land
        self.S_snow = S_next
        non_gl = land & (~glacier)
        P_in = ((P_rain * land) + melt_land) * non_gl
        self.W_land, R_bucket = update_land_bucket(self.W_land, P_in, E_land * non_gl, P, dt)
        self.R_flux = R_bucket + melt_land * glacier
        self.C_snow, self.precip, self.albedo = C_snow, precip, albedo
        self.glacier = glacier
