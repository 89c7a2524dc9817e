"""
qingdai_amd/ocean.py -- device-resident mirror of pygcm/ocean.py:27-561
(WindDrivenSlabOcean): same constructor, `step(dt, u_atm, v_atm, Q_net=None,
ice_mask=None)`, `diagnostics()` and the attributes uo vo eta Ts.  It shares the device
context of the SpectralModel built on the same grid, so the coupled loop stays in HBM.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .device import Device
from .params import QdParams

_ATTR = {"uo": "UO", "vo": "VO", "eta": "ETA", "Ts": "SST"}


class WindDrivenSlabOcean:
    def __init__(self, grid, land_mask, H_m, init_Ts=None, rho_w=None, cp_w=None, params: QdParams | None = None,
                 device=0):
        object.__setattr__(self, "_ready", False)
        dev = getattr(grid, "_device", None)
        if dev is None or dev.h is None:
            dev = Device(grid, params or QdParams.from_env(), device=device)
            dev.upload_now("LAND_MASK", np.ascontiguousarray(land_mask, dtype=np.uint8))
        p = dev.params
        p.H_ocean = float(H_m)
        if rho_w is not None and "QD_RHO_W" not in __import__("os").environ:
            p.rho_w = float(rho_w)
        if cp_w is not None and "QD_CP_W" not in __import__("os").environ:
            p.cp_w = float(cp_w)
        dev.push_params()
        self._dev = dev
        self.grid = grid
        self.land_mask = np.asarray(land_mask, dtype=int)
        self.H = float(H_m)
        self.params = p
        if init_Ts is not None:
            dev.upload_now("SST", np.array(init_Ts, dtype=float))
        self._ready = True

    def __getattr__(self, name):
        if name in _ATTR:
            return self._dev.get(_ATTR[name])
        if name == "_step":
            return self._dev.counters()[1]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__.get("_ready") and name in _ATTR:
            self._dev.set(_ATTR[name], value)
        else:
            object.__setattr__(self, name, value)

    def step(self, dt, u_atm=None, v_atm=None, Q_net=None, ice_mask=None):
        """ocean.py:265-533.  Arrays are uploaded; pass u_atm=None to use the winds resident
        on the device (the SpectralModel's u, v)."""
        dev = self._dev
        if u_atm is not None:
            dev.set("U", u_atm)
        if v_atm is not None:
            dev.set("V", v_atm)
        if Q_net is not None:
            dev.set("QNET", Q_net)
        if ice_mask is not None:
            dev.set("ICE_MASK", np.asarray(ice_mask, dtype=np.uint8))
        if Q_net is None:
            # no heating term: the kernel is told not to use Q_net for this call
            saved = dev.params.ocean_use_qnet
            dev.params.ocean_use_qnet = 0
            dev.push_params()
            dev.ocean_step(dt, 0, ice_mask is not None, 0)
            dev.params.ocean_use_qnet = saved
            dev.push_params()
        else:
            dev.ocean_step(dt, 0, ice_mask is not None, 0)

    def step_coupled(self, dt, inject_sst=True):
        """run_simulation.py:2197-2253 in one call: Q_net and the ice mask are computed on the
        device from the resident atmosphere, then step(), then T_s <- SST over open ocean."""
        self._dev.ocean_step(dt, 1, 1, 1 if inject_sst else 0)

    @property
    def last_n_sub(self):
        return self._dev.last_ocean_nsub()

    def diagnostics(self):
        """ocean.py:535-561"""
        p = self.params
        uo, vo = self.uo, self.vo
        w = np.maximum(np.cos(np.deg2rad(self.grid.lat_mesh)), 0.0)
        wsum = np.sum(w) + 1e-15
        KE = 0.5 * (uo ** 2 + vo ** 2)
        eta = self.eta
        dx_lat = p.a * self.grid.dlat_rad
        dx_lon_min = p.a * self.grid.dlon_rad * max(1e-3, 0.5)
        return {"KE_mean": float(np.sum(KE * w) / wsum), "U_max": float(np.max(np.sqrt(uo ** 2 + vo ** 2))),
                "eta_min": float(np.min(eta)), "eta_max": float(np.max(eta)),
                "cfl_per_s": float(np.sqrt(p.g_ocean * self.H) / max(1e-12, min(dx_lat, dx_lon_min)))}
