"""
oracle/qd_oracle/params.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

Flat parameter namespace for the oracle.  The reference re-reads ~100 QD_*
environment variables inside every step; the oracle takes
them as one explicit object instead so tests can state exactly what they ran.
Field names equal the product's `qingdai_amd.params.QdParams` field names (a
test checks the two default sets agree).  NaN means "env var unset".
"""
from __future__ import annotations

from types import SimpleNamespace

NAN = float("nan")

PLANET_RADIUS = 6.371e6            # constants.py:32
PLANET_OMEGA = 8.726646259971648e-5  # constants.py:34
SIGMA_SB = 5.670374e-8             # constants.py:10


def defaults(**over):
    p = SimpleNamespace(
        # ---- SpectralModel constructor (dynamics.py:22-41; run_simulation.py:1266-1269)
        g=9.81, H=8000.0, tau_rad=10 * 24 * 3600.0, greenhouse_factor=0.40,
        a=PLANET_RADIUS, omega=PLANET_OMEGA,
        seaice_enabled=1, t_freeze=271.35, rho_i=917.0, L_f=3.34e5,
        Cs_ocean=1000.0 * 4200.0 * 50.0, Cs_land=3e6, Cs_ice=5e6,
        q_init_rh=0.5,
        # ---- humidity.py:58-82
        C_E=1.3e-3, rho_a=1.2, h_mbl=800.0, L_v=2.5e6, p0=1.0e5,
        ocean_evap_scale=1.0, land_evap_scale=0.5, ice_evap_scale=0.05,
        tau_cond=1800.0,
        # ---- energy.py:55-74 and the per-step getenv's of dynamics.py:316-386
        sw_a0=0.06, sw_kc=0.20, lw_eps0=0.70, lw_kc=0.20, t_floor=150.0,
        c_sfc=2.0e7,
        energy_w=0.0, cloud_couple=1, rh0=0.6, k_q=0.3, k_p=0.4,
        pcond_ref=NAN, lw_v2=1, hice_ref=0.5, eps_default=0.97,
        ch=1.5e-3, cp_a=1004.0,   # (QD_BOWEN_* only feed a value the path discards: energy.py:444-448)
        atm_h=NAN,  # QD_ATM_H; unset -> h_mbl (dynamics.py:472)
        gh_lock=1, gh_factor_lw=0.582,  # energy.py:122-127 default (driver exports 0.40)
        eps_ocean=0.98, eps_land=0.96, eps_ice=0.99,
        lw_tau0=6.0, lw_ktau=1.0,
        polar_freeze_fix_s=1, polar_freeze_fix_n=1,
        # ---- dynamics.py:484-658
        mom_scheme=0,          # 0 = geos, 1 = primitive
        diff_enable=1,
        filter_type="combo",   # combo | hyper4 | shapiro | spectral
        diff_every=1, sigma4=0.02,
        k4_u=NAN, k4_v=NAN, k4_h=NAN, k4_q=NAN, k4_cloud=NAN,
        k4_nsub=1, diff_q=0, diff_cloud=0,
        shapiro_every=6, shapiro_n=2,
        spec_every=0, spec_cutoff=0.75, spec_damp=0.5,
        diff_factor=0.998,
        # ---- ocean.py:49-75, 380-443, 519-533
        H_ocean=50.0, rho_w=1000.0, cp_w=4200.0, g_ocean=9.81,
        CD=1.5e-3, r_bot=2.0e-5, rho_a_ocean=1.2, vcap=15.0, tau_scale=0.2,
        polar_sponge_lat=70.0, polar_sponge_gain=5.0e-5,
        K_h=5.0e3, sigma4_ocean=0.02, ocean_k4_nsub=1, ocean_diff_every=1,
        ocean_shapiro_n=0, ocean_shapiro_every=8,
        ocean_cfl=0.5, ocean_max_u=3.0, ocean_outlier="mean4",
        ocean_k4_u=NAN, ocean_k4_v=NAN, ocean_k4_eta=NAN,
        ocean_adv_alpha=0.7, ocean_use_qnet=1, ocean_ice_qfac=0.2,
        eta_cap=5.0, ocean_polar_fix=1, ts_min=150.0, ts_max=340.0,
        # ---- driver-side per-step physics (run_simulation.py:1605-1613,1777,1866-1934)
        D_crit=-1e-7, k_precip=1e5, alpha_water=0.1, alpha_ice=0.6, alpha_cloud=0.5,
        p_betadiv=0.4, p_hybrid_fallback=1, pq_min=1e-8, p_blend=0.6,
        pref=NAN, cmax=0.95, w_mem=0.4, w_p=0.4, w_src=0.2,
        cloud_from_p_floor=0.8, cloud_advect=1, cloud_adv_alpha=0.7,
        cloud_smooth_sigma=0.2, use_topo_albedo=1,
        # ---- land hydrology / P019 lapse + snow (hydrology.py:27-80; run_simulation.py:1616-1627)
        runoff_tau_days=10.0, wland_cap_mm=NAN, snow_thresh_K=273.15, snow_melt_rate_mm_day=5.0,
        snow_t_band_K=1.5, snow_ddf_mm_per_k_day=3.0, snow_melt_tref_K=273.15, swe_ref_mm=15.0,
        swe_max_mm=NAN, snow_albedo_fresh=0.70,
        lapse_k_kpm=6.5, land_elev_max_m=10000.0, polar_ice_thick_max_m=4500.0, polar_lat_thresh=60.0,
        rho_snow=300.0, glacier_frac=0.60, glacier_swe_mm=50.0,
        snow_melt_mode=0,      # 0 = degree_day, 1 = constant
        swe_enable=1, lapse_enable=1,
        orog_enable=0, orog_k=7e-4,   # run_simulation.py:1612-1613
        qnet_lw_eps0=NAN, qnet_lw_kc=NAN,   # the driver's autotuned EnergyParams copy (run_simulation.py:2242-2246)
    )
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(f"unknown oracle parameter {k!r}")
        setattr(p, k, v)
    return p


def is_set(x):
    return not (isinstance(x, float) and x != x)
