"""
qingdai_amd/dynamics.py -- device-resident mirror of pygcm/dynamics.py:17-667.

`SpectralModel` keeps the reference's constructor signature, `time_step(Teq, dt,
albedo=None)` and its attribute surface (u v h T_s cloud_cover q h_ice isr isr_A isr_B olr
E_flux_last P_cond_flux_last LH_last LH_release_last cloud_eff_last), so the driver loop of
scripts/run_simulation.py:1760-2490 runs against it unchanged.  State lives in HBM; an
attribute read downloads, an attribute write uploads before the next device step.
"""
from __future__ import annotations

import os

import numpy as np

from .device import Device
from .params import QdParams

_ATTR = {
    "u": "U", "v": "V", "h": "H", "T_s": "TS", "q": "Q", "cloud_cover": "CLOUD", "h_ice": "HICE",
    "isr": "ISR", "isr_A": "ISR_A", "isr_B": "ISR_B", "olr": "OLR",
    "E_flux_last": "EFLUX", "P_cond_flux_last": "PCOND", "LH_last": "LH", "LH_release_last": "LHREL",
    "friction_map": "FRICTION", "C_s_map": "CSMAP",
}


class SpectralModel:
    def __init__(self, grid, friction_map, initial_state=None, g=9.81, H=8000, tau_rad=1e6, greenhouse_factor=0.15,
                 C_s_map=None, land_mask=None, Cs_ocean=None, Cs_land=None, Cs_ice=None, seaice_enabled=None,
                 t_freeze=None, rho_i=None, L_f=None, params: QdParams | None = None, device=0):
        object.__setattr__(self, "_ready", False)
        p = params or QdParams.from_env()
        p.update(g=float(g), H=float(H), tau_rad=float(tau_rad), greenhouse_factor=float(greenhouse_factor))
        if seaice_enabled is not None:
            p.seaice_enabled = int(bool(seaice_enabled))
        for k, v in (("t_freeze", t_freeze), ("rho_i", rho_i), ("L_f", L_f)):
            if v is not None:
                setattr(p, k, float(v))
        # energy.py:393-395 defaults when the caller passes None
        p.Cs_ocean = float(Cs_ocean) if Cs_ocean is not None else 2.0e8
        p.Cs_land = float(Cs_land) if Cs_land is not None else 3.0e6
        p.Cs_ice = float(Cs_ice) if Cs_ice is not None else 5.0e6
        p.has_csmap = 1 if C_s_map is not None else 0
        self.params = p
        self.grid = grid
        self.g, self.H, self.tau_rad, self.greenhouse_factor = p.g, p.H, p.tau_rad, p.greenhouse_factor
        self.a = p.a
        self.dlat_rad, self.dlon_rad = grid.dlat_rad, grid.dlon_rad
        dev = getattr(grid, "_device", None)
        if dev is None or dev.h is None:
            dev = Device(grid, p, device=device)
        else:
            dev.params = p
            dev.push_params()
        self._dev = dev
        if land_mask is None:
            land_mask = np.zeros(grid.lat_mesh.shape, dtype=np.uint8)
        self._land_mask = np.ascontiguousarray(land_mask, dtype=np.uint8)
        dev.upload_now("LAND_MASK", self._land_mask)
        dev.upload_now("FRICTION", friction_map)
        if C_s_map is not None:
            dev.upload_now("CSMAP", C_s_map)
        self.cloud_eff_valid = False
        self.energy_w = p.energy_w
        self._ready = True

    # ---- attribute surface ----------------------------------------------------------
    def __getattr__(self, name):
        if name in _ATTR:
            return self._dev.get(_ATTR[name])
        if name == "cloud_eff_last":
            if not self.__dict__.get("cloud_eff_valid", False):
                raise AttributeError("cloud_eff_last")   # getattr(gcm, 'cloud_eff_last', default) keeps working
            return self._dev.get("CLOUD_EFF")
        if name == "land_mask":
            return self._land_mask
        if name == "_step_counter":
            return self._dev.counters()[0]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__.get("_ready") and name in _ATTR:
            self._dev.set(_ATTR[name], value)
        elif self.__dict__.get("_ready") and name == "cloud_eff_last":
            self._dev.set("CLOUD_EFF", value)
            object.__setattr__(self, "cloud_eff_valid", True)
        elif self.__dict__.get("_ready") and name == "_step_counter":
            self._dev.set_counters(int(value), self._dev.counters()[1])
        elif self.__dict__.get("_ready") and name == "energy_w":
            object.__setattr__(self, name, float(value))
            self.params.energy_w = float(value)
            self._dev.push_params()
        else:
            object.__setattr__(self, name, value)

    def reload_env(self, **over):
        """Re-read the QD_* environment (the reference does so inside every step)."""
        keep = {k: getattr(self.params, k) for k in ("g", "H", "tau_rad", "greenhouse_factor", "Cs_ocean", "Cs_land",
                                                     "Cs_ice", "has_csmap", "H_ocean")}
        p = QdParams.from_env(**keep)
        p.update(**over)
        self.params = p
        self._dev.params = p
        self._dev.push_params()

    # ---- the step ---------------------------------------------------------------------
    def time_step(self, Teq_field, dt, albedo=None):
        """dynamics.py:260-667.  `Teq_field` / `albedo` may be arrays (uploaded) or None to use
        the fields already resident on the device (Device.forcing / simple_albedo)."""
        dev = self._dev
        # The reference re-reads its QD_* variables inside every step (dynamics.py:330-348 and the parameter dataclasses); here
        # they are parsed once (QdParams) because re-parsing ~100 variables costs more than the GPU step.  QD_ENV_REREAD=1 restores
        # the reference's behaviour for callers that change the environment between steps.
        if os.environ.get("QD_ENV_REREAD") == "1":
            self.reload_env()
        if Teq_field is not None:
            dev.set("TEQ", Teq_field)
        if albedo is not None and albedo is not True:
            dev.set("ALBEDO", albedo)
        has_alb = albedo is not None
        dev.atmos_step(dt, has_alb)
        if has_alb:
            object.__setattr__(self, "cloud_eff_valid", True)

    # ---- operator methods kept for parity tests (dynamics.py:90-231) -------------------
    def _laplacian_sphere(self, Fh):
        return self._dev.op_laplacian(Fh)

    def _hyperdiffuse(self, Fh, k4, dt, n_substeps=1):
        return self._dev.op_hyperdiffuse(Fh, k4, dt, n_substeps)

    def _shapiro_filter(self, Fh, n=2, lon_wrap=True):
        return self._dev.op_shapiro(Fh, n)

    def _advect(self, field, dt):
        return self._dev.op_advect(field, self.u, self.v, dt)
