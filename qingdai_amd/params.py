"""
qingdai_amd/params.py -- the QD_* environment surface of the per-step path.

The reference re-reads ~100 environment variables inside every step
(dynamics.py:330-650, ocean.py:380-443, energy.py:122-227,
humidity.py:58-82).  Here they are parsed once into `QdParams` (same names, same
defaults) and sent to the device library as one POD (`qd_params` in
include/qingdai_hip.h).  Call `reload_env()` on a model to pick up changes made to
os.environ mid-run.  NaN means "variable unset".
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, fields

NAN = float("nan")
PLANET_RADIUS = 6.371e6                 # constants.py:32
PLANET_OMEGA = 8.726646259971648e-5     # constants.py:34
SIGMA = 5.670374e-8                     # constants.py:10

_FILTER_CODES = {"combo": 0, "hyper4": 1, "shapiro": 2, "spectral": 3}

# (field, env var, default) -- doubles
_DOUBLES = [
    ("g", None, 9.81), ("H", None, 8000.0), ("tau_rad", None, 10 * 24 * 3600.0), ("greenhouse_factor", None, 0.40),
    ("a", None, PLANET_RADIUS), ("omega", None, PLANET_OMEGA),
    ("t_freeze", "QD_T_FREEZE", 271.35), ("rho_i", "QD_RHO_ICE", 917.0), ("L_f", "QD_LF", 3.34e5),
    ("Cs_ocean", None, 1000.0 * 4200.0 * 50.0), ("Cs_land", "QD_CS_LAND", 3e6), ("Cs_ice", "QD_CS_ICE", 5e6),
    ("C_E", "QD_CE", 1.3e-3), ("rho_a", "QD_RHO_A", 1.2), ("h_mbl", "QD_MBL_H", 800.0), ("L_v", "QD_LV", 2.5e6),
    ("p0", "QD_P0", 1.0e5), ("ocean_evap_scale", "QD_OCEAN_EVAP_SCALE", 1.0),
    ("land_evap_scale", "QD_LAND_EVAP_SCALE", 0.5), ("ice_evap_scale", "QD_ICE_EVAP_SCALE", 0.05),
    ("tau_cond", "QD_TAU_COND", 1800.0),
    ("sw_a0", "QD_SW_A0", 0.06), ("sw_kc", "QD_SW_KC", 0.20), ("lw_eps0", "QD_LW_EPS0", 0.70),
    ("lw_kc", "QD_LW_KC", 0.20), ("t_floor", "QD_T_FLOOR", 150.0), ("c_sfc", "QD_CS", 2.0e7),
    ("energy_w", "QD_ENERGY_W", 0.0), ("rh0", "QD_RH0", 0.6), ("k_q", "QD_K_Q", 0.3), ("k_p", "QD_K_P", 0.4),
    ("pcond_ref", "QD_PCOND_REF", NAN), ("hice_ref", "QD_HICE_REF", 0.5), ("eps_default", "QD_EPS_DEFAULT", 0.97),
    ("ch", "QD_CH", 1.5e-3), ("cp_a", "QD_CP_A", 1004.0),
    ("atm_h", "QD_ATM_H", NAN), ("gh_factor_lw", "QD_GH_FACTOR", 0.582),
    ("eps_ocean", "QD_EPS_OCEAN", 0.98), ("eps_land", "QD_EPS_LAND", 0.96), ("eps_ice", "QD_EPS_ICE", 0.99),
    ("lw_tau0", "QD_LW_TAU0", 6.0), ("lw_ktau", "QD_LW_KTAU", 1.0),
    ("sigma4", "QD_SIGMA4", 0.02), ("k4_u", "QD_K4_U", NAN), ("k4_v", "QD_K4_V", NAN), ("k4_h", "QD_K4_H", NAN),
    ("k4_q", "QD_K4_Q", NAN), ("k4_cloud", "QD_K4_CLOUD", NAN),
    ("spec_cutoff", "QD_SPEC_CUTOFF", 0.75), ("spec_damp", "QD_SPEC_DAMP", 0.5), ("diff_factor", "QD_DIFF_FACTOR", 0.998),
    ("H_ocean", "QD_OCEAN_H_M", 50.0), ("rho_w", "QD_RHO_W", 1000.0), ("cp_w", "QD_CP_W", 4200.0),
    ("g_ocean", None, 9.81), ("CD", "QD_CD", 1.5e-3), ("r_bot", "QD_R_BOT", 2.0e-5), ("rho_a_ocean", "QD_RHO_A", 1.2),
    ("vcap", "QD_WIND_STRESS_VCAP", 15.0), ("tau_scale", "QD_TAU_SCALE", 0.2),
    ("polar_sponge_lat", "QD_POLAR_SPONGE_LAT", 70.0), ("polar_sponge_gain", "QD_POLAR_SPONGE_GAIN", 5.0e-5),
    ("K_h", "QD_KH_OCEAN", 5.0e3), ("sigma4_ocean", "QD_SIGMA4_OCEAN", 0.02), ("ocean_cfl", "QD_OCEAN_CFL", 0.5),
    ("ocean_max_u", "QD_OCEAN_MAX_U", 3.0),
    ("ocean_k4_u", "QD_OCEAN_K4_U", NAN), ("ocean_k4_v", "QD_OCEAN_K4_V", NAN), ("ocean_k4_eta", "QD_OCEAN_K4_ETA", NAN),
    ("ocean_adv_alpha", "QD_OCEAN_ADV_ALPHA", 0.7), ("ocean_ice_qfac", "QD_OCEAN_ICE_QFAC", 0.2),
    ("eta_cap", "QD_ETA_CAP", 5.0), ("ts_min", "QD_TS_MIN", 150.0), ("ts_max", "QD_TS_MAX", 340.0),
    ("D_crit", None, -1e-7), ("k_precip", None, 1e5), ("alpha_water", None, 0.1), ("alpha_ice", None, 0.6),
    ("alpha_cloud", None, 0.5), ("p_betadiv", "QD_P_BETADIV", 0.4), ("pq_min", "QD_PQ_MIN", 1e-8),
    ("p_blend", "QD_P_BLEND", 0.6), ("pref", "QD_PREF", NAN), ("cmax", "QD_CMAX", 0.95),
    ("w_mem", "QD_W_MEM", 0.4), ("w_p", "QD_W_P", 0.4), ("w_src", "QD_W_SRC", 0.2),
    ("cloud_from_p_floor", "QD_CLOUD_FROM_P_FLOOR", 0.8), ("cloud_adv_alpha", "QD_CLOUD_ADV_ALPHA", 0.7),
    ("cloud_smooth_sigma", "QD_CLOUD_SMOOTH_SIGMA", 0.2),
    ("runoff_tau_days", "QD_RUNOFF_TAU_DAYS", 10.0), ("wland_cap_mm", "QD_WLAND_CAP", NAN),
    ("snow_thresh_K", "QD_SNOW_THRESH", 273.15), ("snow_melt_rate_mm_day", "QD_SNOW_MELT_RATE", 5.0),
    ("snow_t_band_K", "QD_SNOW_T_BAND", 1.5), ("snow_ddf_mm_per_k_day", "QD_SNOW_DDF_MM_PER_K_DAY", 3.0),
    ("snow_melt_tref_K", "QD_SNOW_MELT_TREF", 273.15), ("swe_ref_mm", "QD_SWE_REF_MM", 15.0),
    ("swe_max_mm", "QD_SWE_MAX_MM", NAN), ("snow_albedo_fresh", "QD_SNOW_ALBEDO_FRESH", 0.70),
    ("lapse_k_kpm", "QD_LAPSE_K_KPM", 6.5), ("land_elev_max_m", "QD_LAND_ELEV_MAX_M", 10000.0),
    ("polar_ice_thick_max_m", "QD_POLAR_ICE_THICK_MAX_M", 4500.0), ("polar_lat_thresh", "QD_POLAR_LAT_THRESH", 60.0),
    ("rho_snow", "QD_RHO_SNOW", 300.0), ("glacier_frac", "QD_GLACIER_FRAC", 0.60), ("glacier_swe_mm", "QD_GLACIER_SWE_MM", 50.0),
    ("orog_k", "QD_OROG_K", 7e-4),
    ("qnet_lw_eps0", None, NAN), ("qnet_lw_kc", None, NAN),
]
# (field, env var, default) -- int32 switches
_INTS = [
    ("seaice_enabled", "QD_USE_SEAICE", 1), ("cloud_couple", "QD_CLOUD_COUPLE", 1), ("lw_v2", "QD_LW_V2", 1),
    ("gh_lock", "QD_GH_LOCK", 1), ("polar_freeze_fix_s", "QD_POLAR_FREEZE_FIX", 1),
    ("polar_freeze_fix_n", "QD_POLAR_FREEZE_FIX_N", 1),
    ("mom_scheme", "QD_MOM_SCHEME", 0), ("diff_enable", "QD_DIFF_ENABLE", 1), ("filter_type", "QD_FILTER_TYPE", 0),
    ("diff_every", "QD_DIFF_EVERY", 1), ("k4_nsub", "QD_K4_NSUB", 1), ("diff_q", "QD_DIFF_Q", 0),
    ("diff_cloud", "QD_DIFF_CLOUD", 0), ("shapiro_every", "QD_SHAPIRO_EVERY", 6), ("shapiro_n", "QD_SHAPIRO_N", 2),
    ("spec_every", "QD_SPEC_EVERY", 0),
    ("ocean_k4_nsub", "QD_OCEAN_K4_NSUB", 1), ("ocean_diff_every", "QD_OCEAN_DIFF_EVERY", 1),
    ("ocean_shapiro_n", "QD_OCEAN_SHAPIRO_N", 0), ("ocean_shapiro_every", "QD_OCEAN_SHAPIRO_EVERY", 8),
    ("ocean_outlier", "QD_OCEAN_OUTLIER", 0), ("ocean_use_qnet", "QD_OCEAN_USE_QNET", 1),
    ("ocean_polar_fix", "QD_OCEAN_POLAR_FIX", 1),
    ("p_hybrid_fallback", "QD_P_HYBRID_FALLBACK", 1), ("cloud_advect", "QD_CLOUD_ADVECT", 1),
    ("use_topo_albedo", "QD_USE_TOPO_ALBEDO", 1), ("has_csmap", None, 0),
    ("snow_melt_mode", "QD_SNOW_MELT_MODE", 0), ("swe_enable", "QD_SWE_ENABLE", 1), ("lapse_enable", "QD_LAPSE_ENABLE", 1),
    ("orog_enable", "QD_OROG", 0),
]


class qd_params(ctypes.Structure):
    """ctypes mirror of `struct qd_params` (include/qingdai_hip.h) -- same order."""
    _fields_ = [(n, ctypes.c_double) for n, _, _ in _DOUBLES] + [(n, ctypes.c_int32) for n, _, _ in _INTS]


def _env_float(name, default):
    v = os.environ.get(name)
    if v is None or v == "" or v in ("None", "none", "null"):
        return default
    try:
        return float(v)
    except Exception:
        return default


def _env_int(name, default):
    v = os.environ.get(name)
    if v is None or v == "":
        return default
    if name == "QD_MOM_SCHEME":
        return 1 if v.strip().lower() == "primitive" else 0
    if name == "QD_FILTER_TYPE":
        return _FILTER_CODES.get(v.strip().lower(), 4)
    if name == "QD_OCEAN_OUTLIER":
        return 0 if v.strip().lower() == "mean4" else 1
    if name == "QD_SNOW_MELT_MODE":
        return 0 if v.strip().lower() == "degree_day" else 1
    try:
        return int(v)
    except Exception:
        return default


class QdParams:
    """Flat parameter set; attribute names == qd_params fields == oracle parameter names."""

    def __init__(self, **over):
        for n, _, d in _DOUBLES:
            setattr(self, n, float(d))
        for n, _, d in _INTS:
            setattr(self, n, int(d))
        self.q_init_rh = 0.5
        self.update(**over)

    def update(self, **over):
        for k, v in over.items():
            if k == "filter_type" and isinstance(v, str):
                v = _FILTER_CODES.get(v.lower(), 4)
            if k == "ocean_outlier" and isinstance(v, str):
                v = 0 if v.lower() == "mean4" else 1
            if not hasattr(self, k):
                raise AttributeError(f"unknown parameter {k!r}")
            setattr(self, k, v)
        return self

    @classmethod
    def from_env(cls, **over):
        p = cls()
        for n, env, d in _DOUBLES:
            if env is not None:
                setattr(p, n, _env_float(env, d))
        for n, env, d in _INTS:
            if env is not None:
                setattr(p, n, _env_int(env, d))
        p.q_init_rh = _env_float("QD_Q_INIT_RH", 0.5)
        p.update(**over)
        return p

    def to_struct(self):
        s = qd_params()
        for n, _, _ in _DOUBLES:
            setattr(s, n, float(getattr(self, n)))
        for n, _, _ in _INTS:
            setattr(s, n, int(getattr(self, n)))
        return s

    def as_oracle_kwargs(self):
        """Same values under the oracle's names (filter_type / ocean_outlier as strings)."""
        inv = {v: k for k, v in _FILTER_CODES.items()}
        d = {n: getattr(self, n) for n, _, _ in _DOUBLES}
        d.update({n: getattr(self, n) for n, _, _ in _INTS if n not in ("has_csmap",)})
        d["filter_type"] = inv.get(self.filter_type, "other")
        d["ocean_outlier"] = "mean4" if self.ocean_outlier == 0 else "clamp"
        d["q_init_rh"] = self.q_init_rh
        return d
