"""
oracle/qd_oracle/ocean.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

NumPy restatement of WindDrivenSlabOcean (pygcm/ocean.py:27-561): wind stress,
CFL-derived sub-step count, momentum / continuity / SST advection-diffusion /
Q_net heating / outlier handling per sub-step, polar ring averaging, clamps.
"""
from __future__ import annotations

import numpy as np

from . import numerics as nx
from .atmos import laplacian_sphere, hyperdiffuse, advect_semilag
from .params import is_set


class OceanOracle:
    def __init__(self, grid, land_mask, P, init_Ts=None):
        self.grid = grid
        self.P = P
        self.land_mask = np.asarray(land_mask, dtype=int)
        self.H = float(P.H_ocean)
        self.a = P.a
        self.dlat = grid.dlat_rad
        self.dlon = grid.dlon_rad
        self.lat_rad = np.deg2rad(grid.lat_mesh)
        self.coslat = np.maximum(np.cos(self.lat_rad), 0.5)  # ocean.py:82
        self.f = grid.coriolis_param
        shape = grid.lat_mesh.shape
        self.uo = np.zeros(shape)
        self.vo = np.zeros(shape)
        self.eta = np.zeros(shape)
        self.Ts = np.full(shape, 288.0) if init_Ts is None else np.array(init_Ts, dtype=float, copy=True)
        self._step = 0
        self.last_n_sub = 0

    # ---- ocean.py:197-262
    def _polar_scalar_fill(self, F, ocean_mask):
        for j in (0, -1):
            m = ocean_mask[j, :]
            if np.any(m):
                F[j, m] = float(np.mean(F[j, m]))

    def _polar_vector_fill(self, u, v, ocean_mask):
        lam = np.deg2rad(self.grid.lon)

        def basis(l, north):
            ee = np.stack([-np.sin(l), np.cos(l), np.zeros_like(l)], axis=1)
            if north:
                en = np.stack([-np.cos(l), -np.sin(l), np.zeros_like(l)], axis=1)
            else:
                en = np.stack([np.cos(l), np.sin(l), np.zeros_like(l)], axis=1)
            return ee, en
        for j, north in ((0, False), (-1, True)):
            m = ocean_mask[j, :]
            if not np.any(m):
                continue
            idx = np.where(m)[0]
            ee, en = basis(lam[idx], north)
            v3 = ee * u[j, idx][:, None] + en * v[j, idx][:, None]
            v3m = np.mean(v3, axis=0)
            ea, na = basis(lam, north)
            uf = ea @ v3m
            vf = na @ v3m
            u[j, m] = uf[m]
            v[j, m] = vf[m]

    def n_substeps(self, dt, Va):
        """ocean.py:293-303"""
        P = self.P
        dx_lat = self.a * self.dlat
        dx_lon_min = self.a * self.dlon * max(1e-3, float(np.min(self.coslat)))
        dx_min = min(dx_lat, dx_lon_min)
        c = np.sqrt(P.g_ocean * self.H)
        uadv = float(np.max(np.sqrt(self.uo ** 2 + self.vo ** 2)))
        uadv = max(uadv, float(np.max(Va)))
        target = max(1e-3, P.ocean_cfl)
        n_sub = int(np.ceil(max(c, uadv) * (dt / max(1e-12, dx_min)) / target))
        return int(max(1, min(500, n_sub)))

    def step(self, dt, u_atm, v_atm, Q_net=None, ice_mask=None):
        """ocean.py:265-533"""
        P = self.P
        g = P.g_ocean
        self._step += 1
        u_rel = u_atm - self.uo
        v_rel = v_atm - self.vo
        Va = np.sqrt(u_rel ** 2 + v_rel ** 2)
        Va_eff = np.minimum(Va, P.vcap)
        tau_x = P.tau_scale * (P.rho_a_ocean * P.CD * Va_eff * u_rel)
        tau_y = P.tau_scale * (P.rho_a_ocean * P.CD * Va_eff * v_rel)
        n_sub = self.n_substeps(dt, Va)
        self.last_n_sub = n_sub
        sub_dt = dt / n_sub
        on_land = (self.land_mask == 1)
        ocean_mask = (self.land_mask == 0)

        for _ in range(n_sub):
            deta_dlam = (np.roll(self.eta, -1, axis=1) - np.roll(self.eta, 1, axis=1)) / (2.0 * self.dlon)
            deta_dphi = (np.roll(self.eta, -1, axis=0) - np.roll(self.eta, 1, axis=0)) / (2.0 * self.dlat)
            gx = deta_dlam / (self.a * self.coslat)
            gy = deta_dphi / self.a
            du = (self.f * self.vo - g * gx + tau_x / (P.rho_w * self.H) - P.r_bot * self.uo)
            dv = (-self.f * self.uo - g * gy + tau_y / (P.rho_w * self.H) - P.r_bot * self.vo)
            self.uo = self.uo + sub_dt * du
            self.vo = self.vo + sub_dt * dv
            self.uo[on_land] = 0.0
            self.vo[on_land] = 0.0
            # polar sponge (ocean.py:331-336)
            lat_deg = np.abs(np.rad2deg(self.lat_rad))
            s = np.clip((lat_deg - P.polar_sponge_lat) / max(1e-6, 90.0 - P.polar_sponge_lat), 0.0, 1.0)
            r_extra = P.polar_sponge_gain * (s ** 2)
            self.uo = self.uo - sub_dt * r_extra * self.uo
            self.vo = self.vo - sub_dt * r_extra * self.vo
            # del^4 (ocean.py:341-356)
            if (P.ocean_diff_every > 0) and (self._step % int(P.ocean_diff_every) == 0):
                dx_min_map = np.minimum(self.a * self.dlat, self.a * self.dlon * self.coslat)
                k4_map = P.sigma4_ocean * (dx_min_map ** 4) / max(1e-12, sub_dt)
                k4_u = float(P.ocean_k4_u) if is_set(P.ocean_k4_u) else k4_map
                k4_v = float(P.ocean_k4_v) if is_set(P.ocean_k4_v) else k4_map
                k4_e = float(P.ocean_k4_eta) if is_set(P.ocean_k4_eta) else 0.5 * k4_map
                ns = int(P.ocean_k4_nsub)
                self.uo = hyperdiffuse(self.uo, k4_u, sub_dt, ns, self.dlat, self.dlon, self.coslat, self.a)
                self.vo = hyperdiffuse(self.vo, k4_v, sub_dt, ns, self.dlat, self.dlon, self.coslat, self.a)
                self.eta = hyperdiffuse(self.eta, k4_e, sub_dt, ns, self.dlat, self.dlon, self.coslat, self.a)
            if (P.ocean_shapiro_n > 0) and (P.ocean_shapiro_every > 0) and (self._step % int(P.ocean_shapiro_every) == 0):
                self.uo = nx.shapiro(self.uo, P.ocean_shapiro_n)
                self.vo = nx.shapiro(self.vo, P.ocean_shapiro_n)
                self.eta = nx.shapiro(self.eta, P.ocean_shapiro_n)
            # continuity (ocean.py:365-377)
            div = self.grid.divergence(self.uo, self.vo)
            self.eta = self.eta + (-sub_dt * self.H * div)
            self.eta[on_land] = 0.0
            if np.any(ocean_mask):
                w = np.maximum(np.cos(self.lat_rad), 0.0)
                w_o = w * ocean_mask
                eta_mean = float(np.sum(self.eta * w_o) / (np.sum(w_o) + 1e-15))
                self.eta = self.eta - eta_mean
            # SST advection + lateral diffusion (ocean.py:380-386)
            al = float(P.ocean_adv_alpha)
            Ts_adv = advect_semilag(self.Ts, self.uo, self.vo, sub_dt, self.a, self.dlat, self.dlon, self.coslat)
            self.Ts = (1.0 - al) * self.Ts + al * Ts_adv
            if P.K_h > 0.0:
                self.Ts = self.Ts + sub_dt * P.K_h * laplacian_sphere(self.Ts, self.dlat, self.dlon, self.coslat, self.a)
            # vertical heat flux (ocean.py:389-406)
            if P.ocean_use_qnet and (Q_net is not None):
                heat = Q_net / (P.rho_w * P.cp_w * self.H)
                if ice_mask is not None:
                    open_m = ocean_mask & (~ice_mask)
                    ice_m = ocean_mask & ice_mask
                    Tn = np.where(open_m, self.Ts + sub_dt * heat, self.Ts)
                    if P.ocean_ice_qfac > 0.0:
                        Tn = np.where(ice_m, Tn + sub_dt * P.ocean_ice_qfac * heat, Tn)
                    self.Ts = Tn
                else:
                    self.Ts = np.where(ocean_mask, self.Ts + sub_dt * heat, self.Ts)
            # outliers (ocean.py:409-434)
            self.uo = np.nan_to_num(self.uo)
            self.vo = np.nan_to_num(self.vo)
            speed = np.sqrt(self.uo ** 2 + self.vo ** 2)
            cap = float(P.ocean_max_u)
            if P.ocean_outlier == "mean4":
                um = 0.25 * (np.roll(self.uo, -1, 0) + np.roll(self.uo, 1, 0) + np.roll(self.uo, -1, 1) + np.roll(self.uo, 1, 1))
                vm = 0.25 * (np.roll(self.vo, -1, 0) + np.roll(self.vo, 1, 0) + np.roll(self.vo, -1, 1) + np.roll(self.vo, 1, 1))
                fast = speed > cap
                self.uo = np.where(fast, um, self.uo)
                self.vo = np.where(fast, vm, self.vo)
                sp2 = np.sqrt(self.uo ** 2 + self.vo ** 2)
                sc2 = np.where(sp2 > cap, cap / (sp2 + 1e-12), 1.0)
                self.uo = self.uo * sc2
                self.vo = self.vo * sc2
            else:
                sc = np.where(speed > cap, cap / (speed + 1e-12), 1.0)
                self.uo = self.uo * sc
                self.vo = self.vo * sc
            self.eta = np.clip(np.nan_to_num(self.eta), -P.eta_cap, P.eta_cap)
            self.Ts = np.nan_to_num(self.Ts)

        if P.ocean_polar_fix:
            self._polar_scalar_fill(self.Ts, ocean_mask)
            self._polar_vector_fill(self.uo, self.vo, ocean_mask)
        self.Ts = np.clip(self.Ts, P.ts_min, P.ts_max)

    def diagnostics(self):
        """ocean.py:535-561"""
        P = self.P
        w = np.maximum(np.cos(self.lat_rad), 0.0)
        wsum = np.sum(w) + 1e-15
        KE = 0.5 * (self.uo ** 2 + self.vo ** 2)
        dx_lat = self.a * self.dlat
        dx_lon_min = self.a * self.dlon * max(1e-3, float(np.min(self.coslat)))
        return {"KE_mean": float(np.sum(KE * w) / wsum),
                "U_max": float(np.max(np.sqrt(self.uo ** 2 + self.vo ** 2))),
                "eta_min": float(np.min(self.eta)), "eta_max": float(np.max(self.eta)),
                "cfl_per_s": float(np.sqrt(P.g_ocean * self.H) / max(1e-12, min(dx_lat, dx_lon_min)))}
