"""
oracle/qd_oracle/atmos.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

NumPy restatement of SpectralModel (pygcm/dynamics.py:17-667): state layout,
the lat-lon operators (spherical Laplacian, del^4 hyperdiffusion, Shapiro,
zonal-FFT filter, semi-Lagrangian advection) and the whole `time_step`.
Parameters come from the flat namespace of params.py instead of os.getenv.
"""
from __future__ import annotations

import numpy as np

from . import column as col
from . import numerics as nx
from .params import SIGMA_SB, is_set


# ---------------------------------------------------------------- operators
def laplacian_sphere(F, dlat, dlon, coslat, a):
    """dynamics.py:144-173 / ocean.py:100-117 / jax_compat.py:111-132.
    `coslat` is the caller's already-floored cos(phi) map (0.2 atmos, 0.5 ocean).
    Like the reference (`np.nan_to_num(F, copy=False)`, dynamics.py:165 / ocean.py:112) this scrubs the CALLER's array in
    place -- a side effect its callers rely on: `self.Ts += dt*K_h*lap(self.Ts)` (ocean.py:386) adds to the scrubbed Ts, and
    the second sub-step of _hyperdiffuse continues from a scrubbed `out`."""
    F = np.nan_to_num(F, copy=False)
    dF = nx.gradient_axis0(F, dlat)
    term_phi = (1.0 / coslat) * nx.gradient_axis0(coslat * dF, dlat)
    d2 = (np.roll(F, -1, axis=1) - 2.0 * F + np.roll(F, 1, axis=1)) / (dlon ** 2)
    term_lam = d2 / (coslat ** 2)
    return (term_phi + term_lam) / (a ** 2)


def hyperdiffuse(F, k4, dt, n_substeps, dlat, dlon, coslat, a):
    """dynamics.py:175-212 / ocean.py:119-152 / jax_compat.py:135-187."""
    if dt <= 0.0:
        return F
    if np.isscalar(k4):
        k4a = float(k4)
        if k4a <= 0.0:
            return F
    else:
        k4a = np.nan_to_num(k4)
        if np.all(k4a <= 0.0):
            return F
    n = max(1, int(n_substeps))
    sub_dt = dt / n
    out = np.nan_to_num(F, copy=True)
    for _ in range(n):
        L = laplacian_sphere(out, dlat, dlon, coslat, a)
        L2 = laplacian_sphere(L, dlat, dlon, coslat, a)
        out = out - k4a * L2 * sub_dt
    return np.nan_to_num(out)


def spectral_zonal_filter(F, cutoff, damp, n_lon):
    """dynamics.py:233-258"""
    if damp <= 0.0 or cutoff <= 0.0:
        return np.nan_to_num(F)
    arr = np.nan_to_num(F)
    fft = np.fft.rfft(arr, axis=1)
    bins = fft.shape[1]
    if bins <= 1:
        return arr
    kN = bins - 1
    kcut = int(max(1, min(kN, int(cutoff * kN))))
    factor = np.ones(bins, dtype=float)
    factor[kcut:] *= max(0.0, 1.0 - min(1.0, damp))
    fft = fft * factor[np.newaxis, :]
    return np.nan_to_num(np.fft.irfft(fft, n=n_lon, axis=1))


def advect_semilag(field, u, v, dt, a, dlat, dlon, coslat):
    """dynamics.py:90-118 / ocean.py:166-194 / run_simulation.py:1131-1158 /
    jax_compat.py:190-216.  `coslat` is the caller's floored map (1e-6 / 0.5)."""
    dlam = u * dt / (a * coslat)
    dphi = v * dt / a
    dx = dlam / dlon
    dy = dphi / dlat
    nlat, nlon = field.shape
    JJ, II = np.meshgrid(np.arange(nlat), np.arange(nlon), indexing="ij")
    return nx.bilinear_wrap(field, JJ - dy, II - dx)


# ---------------------------------------------------------------- the model
class AtmosOracle:
    """State + time_step of the reference SpectralModel (dynamics.py:22-88, 260-667)."""

    def __init__(self, grid, friction_map, land_mask, P, C_s_map=None):
        self.grid = grid
        self.P = P
        self.friction_map = np.asarray(friction_map, dtype=float)
        self.land_mask = land_mask
        self.C_s_map = C_s_map
        shape = grid.lat_mesh.shape
        lat_rad = np.deg2rad(grid.lat_mesh)
        self.u = np.zeros(shape)
        self.v = np.zeros(shape)
        self.h = np.full(shape, P.H, dtype=float) + 300 * (np.sin(lat_rad) ** 2)
        self.T_s = np.full(shape, 288.0)
        self.cloud_cover = np.zeros(shape)
        self.h_ice = np.zeros(shape)
        self.isr = np.zeros(shape)
        self.isr_A = np.zeros(shape)
        self.isr_B = np.zeros(shape)
        self.olr = np.zeros(shape)
        self.q = col.q_init(self.T_s, P.q_init_rh, P.p0)
        self.E_flux_last = np.zeros(shape)
        self.P_cond_flux_last = np.zeros(shape)
        self.LH_last = np.zeros(shape)
        self.LH_release_last = np.zeros(shape)
        self.cloud_eff_last = None
        self._step_counter = 0
        self.dlat = grid.dlat_rad
        self.dlon = grid.dlon_rad
        self._cos = np.cos(lat_rad)

    # -- operator wrappers with the atmosphere's cos floors
    def _advect(self, field, dt):
        return advect_semilag(field, self.u, self.v, dt, self.P.a, self.dlat, self.dlon,
                              np.maximum(1e-6, self._cos))

    def _hyper(self, F, k4, dt, nsub=1):
        return hyperdiffuse(F, k4, dt, nsub, self.dlat, self.dlon,
                            np.maximum(self._cos, 0.2), self.P.a)

    def time_step(self, Teq, dt, albedo=None):
        P = self.P
        g = P.g
        T_a = 288.0 + (g / 1004.0) * self.h

        # --- humidity column (dynamics.py:282-297)
        fac = col.surface_evaporation_factor(self.land_mask, self.h_ice, P)
        E = col.evaporation_flux(self.T_s, self.q, self.u, self.v, fac, P)
        LH = P.L_v * E
        M_col = max(1e-6, float(P.rho_a * P.h_mbl))
        q_evap = self.q + (E / M_col) * dt
        P_cond, q_after = col.condensation(q_evap, T_a, dt, P)
        LH_release = P.L_v * P_cond
        self.q = np.clip(np.nan_to_num(q_after), 0.0, 0.5)
        self.E_flux_last = E
        self.P_cond_flux_last = P_cond
        self.LH_last = LH
        self.LH_release_last = LH_release

        # --- Newton path (dynamics.py:304-322)
        olr_old = SIGMA_SB * self.T_s ** 4
        net_old = SIGMA_SB * Teq ** 4 + P.greenhouse_factor * SIGMA_SB * T_a ** 4 - olr_old
        Ts_newton = self.T_s + (net_old / max(1e-12, P.c_sfc)) * dt

        Ts_energy = None
        h_ice_next = None
        if albedo is not None:
            # --- cloud optical consistency (dynamics.py:329-353)
            if P.cloud_couple:
                qsat_air = col.q_sat(T_a, P.p0)
                RH = np.clip(self.q / np.maximum(1e-12, qsat_air), 0.0, 1.5)
                rh_excess = np.maximum(0.0, RH - P.rh0)
                Pc = self.P_cond_flux_last
                if is_set(P.pcond_ref):
                    P_ref = float(P.pcond_ref)
                else:
                    P_ref = nx.median_positive(Pc, 1e-6)
                p_term = np.tanh(Pc / P_ref) if P_ref > 0 else np.tanh(np.zeros_like(Pc))
                cloud_eff = np.clip(self.cloud_cover + P.k_q * rh_excess + P.k_p * p_term, 0.0, 1.0)
            else:
                cloud_eff = self.cloud_cover
            self.cloud_eff_last = cloud_eff
            SW_atm, SW_sfc, R = col.shortwave(self.isr, albedo, cloud_eff, P)
            if P.lw_v2:
                ice_frac = 1.0 - np.exp(-np.maximum(self.h_ice, 0.0) / max(1e-6, P.hice_ref))
                eps_map = col.surface_emissivity_map(self.land_mask, ice_frac, P)
                LW_atm, LW_sfc, OLR, DLR, eps = col.longwave_v2(self.T_s, T_a, cloud_eff, eps_map, P)
            else:
                LW_atm, LW_sfc, OLR, DLR, eps = col.longwave_v1(self.T_s, T_a, cloud_eff, P)
            SH = col.sensible_heat(self.T_s, T_a, self.u, self.v, P)
            if P.seaice_enabled:
                Ts_energy, h_ice_next = col.integrate_surface_energy_with_seaice(
                    self.T_s, SW_sfc, LW_sfc, SH, LH, dt, self.land_mask, self.h_ice, P)
            elif self.C_s_map is not None:
                Ts_energy = col.integrate_surface_energy_map(self.T_s, SW_sfc, LW_sfc, SH, LH, dt, self.C_s_map, P)
            else:
                Ts_energy = col.integrate_surface_energy(self.T_s, SW_sfc, LW_sfc, SH, LH, dt, P)
            self.olr = OLR
            self._last_fluxes = dict(SW_atm=SW_atm, SW_sfc=SW_sfc, R=R, LW_atm=LW_atm,
                                     LW_sfc=LW_sfc, OLR=OLR, DLR=DLR, SH=SH, LH=LH)
        else:
            self.olr = olr_old

        w = min(1.0, max(0.0, float(P.energy_w)))
        if Ts_energy is None:
            self.T_s = Ts_newton
        else:
            self.T_s = (1.0 - w) * Ts_newton + w * Ts_energy
            if P.seaice_enabled and h_ice_next is not None:
                self.h_ice = h_ice_next
        self._step_counter += 1

        # --- gentle semi-Lagrangian advection of T_s and q (dynamics.py:454-461)
        al = 0.2
        self.T_s = (1.0 - al) * self.T_s + al * self._advect(self.T_s, dt)
        self.q = (1.0 - al) * self.q + al * self._advect(self.q, dt)
        self.q = np.clip(np.nan_to_num(self.q), 0.0, 0.5)

        # --- radiative relaxation of h (dynamics.py:464-467)
        h_eq = (287 / g) * Teq
        self.h = self.h + ((h_eq - self.h) / P.tau_rad) * dt
        # --- atmospheric energy -> h (dynamics.py:470-480)
        if (albedo is not None) and (P.energy_w > 0.0):
            H_atm = P.atm_h if is_set(P.atm_h) else P.h_mbl
            self.h = col.integrate_atmos_energy_height(self.h, SW_atm, LW_atm, SH, LH_release, dt,
                                                       P.rho_a, H_atm, g, float(P.energy_w))

        # --- momentum (dynamics.py:482-530)
        f = self.grid.coriolis_param
        dh_dlon = nx.gradient_axis1(self.h, self.dlon)
        dh_dlat = nx.gradient_axis0(self.h, self.dlat)
        cosc = np.maximum(self._cos, 1e-6)
        if P.mom_scheme == 1:
            u_old = self.u.copy()
            v_old = self.v.copy()
            PGF_x = -(g / (P.a * cosc)) * dh_dlon
            PGF_y = -(g / P.a) * dh_dlat
            du = (PGF_x + f * v_old - self.friction_map * u_old) * dt
            dv = (PGF_y - f * u_old - self.friction_map * v_old) * dt
            self.u = np.clip(u_old + du, -200.0, 200.0)
            self.v = np.clip(v_old + dv, -200.0, 200.0)
        else:
            f_min = 2.0 * P.omega * np.sin(np.deg2rad(5.0))
            sgn = np.where(f >= 0.0, 1.0, -1.0)
            f_safe = np.where(np.abs(f) < f_min, sgn * f_min, f)
            u_g = np.clip(-(g / (f_safe * P.a * cosc)) * dh_dlat, -200.0, 200.0)
            v_g = np.clip((g / (f_safe * P.a)) * dh_dlon, -200.0, 200.0)
            self.u = self.u * 0.8 + u_g * 0.2
            self.v = self.v * 0.8 + v_g * 0.2
            self.u = self.u + (-self.friction_map * self.u) * dt
            self.v = self.v + (-self.friction_map * self.v) * dt

        # --- del^4 hyperdiffusion (dynamics.py:533-594)
        ftype = str(P.filter_type).lower()
        sc = self._step_counter
        if P.diff_enable and ftype in ("hyper4", "combo") and (sc % max(1, int(P.diff_every)) == 0):
            cos3 = np.maximum(self._cos, 1e-3)
            dx_min = np.minimum(P.a * self.dlat, P.a * self.dlon * cos3)
            k4_base = P.sigma4 * (dx_min ** 4) / max(1e-12, dt)
            k4_u = float(P.k4_u) if is_set(P.k4_u) else k4_base
            k4_v = float(P.k4_v) if is_set(P.k4_v) else k4_base
            k4_h = float(P.k4_h) if is_set(P.k4_h) else 0.5 * k4_base
            k4_q = float(P.k4_q) if is_set(P.k4_q) else 0.5 * k4_base
            k4_c = float(P.k4_cloud) if is_set(P.k4_cloud) else 0.25 * k4_base
            ns = int(P.k4_nsub)
            self.u = self._hyper(self.u, k4_u, dt, ns)
            self.v = self._hyper(self.v, k4_v, dt, ns)
            self.h = self._hyper(self.h, k4_h, dt, ns)

            def _pos(k):
                return (np.isscalar(k) and k > 0.0) or ((not np.isscalar(k)) and bool(np.any(k > 0.0)))
            if _pos(k4_q) or P.diff_q:
                self.q = self._hyper(self.q, k4_q, dt)
            if _pos(k4_c) or P.diff_cloud:
                self.cloud_cover = self._hyper(self.cloud_cover, k4_c, dt)

        # --- Shapiro / spectral (dynamics.py:610-637)
        she = int(P.shapiro_every)
        if ftype in ("shapiro", "combo", "hyper4") and she > 0 and (sc % she == 0):
            n = int(P.shapiro_n)
            self.u = nx.shapiro(self.u, n)
            self.v = nx.shapiro(self.v, n)
            self.h = nx.shapiro(self.h, n)
            if P.diff_q:
                self.q = nx.shapiro(self.q, max(1, n - 1))
            if P.diff_cloud:
                self.cloud_cover = nx.shapiro(self.cloud_cover, max(1, n - 1))
        spe = int(P.spec_every)
        if ftype in ("spectral", "combo") and spe > 0 and (sc % spe == 0):
            nl = self.grid.n_lon
            self.u = spectral_zonal_filter(self.u, P.spec_cutoff, P.spec_damp, nl)
            self.v = spectral_zonal_filter(self.v, P.spec_cutoff, P.spec_damp, nl)
            self.h = spectral_zonal_filter(self.h, P.spec_cutoff, P.spec_damp, nl)

        # --- cloud advect / decay / global damp / scrub (dynamics.py:642-667)
        self.cloud_cover = self._advect(self.cloud_cover, dt)
        self.cloud_cover = self.cloud_cover * (1 - dt / (2.0 * 24 * 3600))
        d = float(P.diff_factor)
        self.u = np.nan_to_num(self.u * d)
        self.v = np.nan_to_num(self.v * d)
        self.h = np.nan_to_num(self.h * d)
        self.cloud_cover = np.nan_to_num(self.cloud_cover * d)
        self.q = np.nan_to_num(self.q * d)
        self.T_s = np.nan_to_num(self.T_s)
