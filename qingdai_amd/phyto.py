"""
qingdai_amd/phyto.py -- transport of the phytoplankton tracers by the ocean currents
(pygcm/ecology/phyto.py:496-547) on the MI355X library.

`PhytoManager.advect_diffuse` reuses the ocean's two operators per species -- the semi-Lagrangian gather
and the spherical Laplacian, both on the ocean cos floor max(cos, 0.5).

Two forms:
  * `PhytoTracers` -- the tracers RESIDENT on the device (qd_phyto_*, csrc/qd_phyto.hip): all species in three launches per step,
    inside the resident loop of qd_step_n (flags bit6) on the currents the ocean step has just written, the way the reference
    driver calls it at scripts/run_simulation.py:2254-2258.  `driver.Simulation` creates one under QD_PHYTO_ENABLE /
    QD_PHYTO_ADVECTION (both default 1, run_simulation.py:1347,1351).
  * `advect_diffuse(dev, C_s, uo, vo, ...)` -- the operator-seam form for a host ecology that owns the [S, n_lat, n_lon] array:
    `qd_op_advect` + `qd_op_laplacian` (cos kind 1) per species, blend / clip / land mask / polar means in NumPy.
The ecology that feeds on the tracers (daily growth, optics, genes) stays outside this path.
"""
from __future__ import annotations

import os

import numpy as np


def _env_list(name):
    v = os.getenv(name)
    if not v:
        return []
    try:
        return [float(x) for x in v.replace(";", ",").split(",") if x.strip()]
    except ValueError:
        return []


class PhytoTracers:
    """The prognostic part of the reference's PhytoManager that the per-step path touches: C_phyto_s [S, n_lat, n_lon] (mg Chl / m^3),
    initialised like phyto.py:152-158,253-270 (S = QD_PHYTO_NSPECIES (10), equal or QD_PHYTO_INIT_FRAC fractions of QD_PHYTO_CHL0
    (0.05) over the ocean, 0 on land), K_h = QD_PHYTO_KH | QD_KH_OCEAN (5e3, phyto.py:123), alpha = QD_PHYTO_ADV_ALPHA (0.7).
    The array lives on the device; `C_phyto_s` downloads / uploads it."""

    def __init__(self, grid, land_mask, dev=None):
        self.grid = grid
        self.land_mask = np.asarray(land_mask)
        self.ocean_mask = (self.land_mask == 0)
        try:
            self.S = max(1, int(os.getenv("QD_PHYTO_NSPECIES", "10")))
        except ValueError:
            self.S = 10
        self.K_h = float(os.getenv("QD_PHYTO_KH", os.getenv("QD_KH_OCEAN", "5.0e3")))
        self.adv_alpha = float(os.getenv("QD_PHYTO_ADV_ALPHA", "0.7"))
        self.chl0 = float(os.getenv("QD_PHYTO_CHL0", "0.05"))
        frac = _env_list("QD_PHYTO_INIT_FRAC")
        if len(frac) >= self.S:
            f = np.clip(np.array(frac[:self.S], dtype=float), 0.0, None)
            tot = float(np.sum(f))
            f = f / tot if tot > 0 else np.full((self.S,), 1.0 / self.S)
        else:
            f = np.full((self.S,), 1.0 / self.S)
        self.init_frac_s = f
        self.dev = None
        self._host = self.default_state()
        if dev is not None:
            self.attach(dev)

    def default_state(self):
        C = np.zeros((self.S, self.grid.n_lat, self.grid.n_lon))
        for s in range(self.S):
            C[s] = self.init_frac_s[s] * self.chl0
            C[s, ~self.ocean_mask] = 0.0
        return C

    def attach(self, dev):
        self.dev = dev
        dev.phyto_configure(self.S, self.K_h, self.adv_alpha)
        dev.phyto_upload(self._host)
        self._host = None

    @property
    def C_phyto_s(self):
        return self.dev.phyto_download() if self.dev is not None else self._host

    @C_phyto_s.setter
    def C_phyto_s(self, C):
        C = np.clip(np.asarray(C, dtype=np.float64), 0.0, np.inf)          # load_autosave: clip, land = 0 (phyto.py:625-628)
        C[:, ~self.ocean_mask] = 0.0
        if self.dev is not None:
            self.dev.phyto_upload(C)
        else:
            self._host = C

    def advect_diffuse(self, dt_seconds):
        """One transport step on the resident currents (outside qd_step_n; inside it: step_n(..., phyto=True))."""
        self.dev.phyto_advect_diffuse(dt_seconds)

    # -- data/plankton.nc, the tracer part of save_distribution_nc / load_distribution_nc (phyto.py:737-802)
    def save_distribution_nc(self, path, day_value=None):
        """phyto.py:737-802.  This build owns the tracer part of plankton.nc (lat, lon, C_phyto_s, S, day).  The reference's file also
        holds what its DAILY host code maintains (alpha_water_scalar, Kd_490, the prognostic nutrient pool N, alpha_water_bands,
        bands_lambda_centers, the band dimension, H_mld_m, NB): when a file of the same grid and species count is already there --
        a data directory shared with a reference run -- those variables and attributes are carried through the rewrite."""
        from . import ncio
        try:
            g = self.grid
            dims = {"lat": g.n_lat, "lon": g.n_lon, "species": self.S}
            v = {"lat": ("f4", ("lat",), np.asarray(g.lat, np.float32)), "lon": ("f4", ("lon",), np.asarray(g.lon, np.float32)),
                 "C_phyto_s": ("f4", ("species", "lat", "lon"), self.C_phyto_s.astype(np.float32))}
            attrs = {"title": "Qingdai Phytoplankton Distributions", "S": int(self.S)}
            if os.path.exists(path):
                try:
                    odims, ovars, oattrs = ncio.read_nc_full(path)
                    if all(odims.get(k) == n for k, n in dims.items()):
                        for d, n in odims.items():
                            dims.setdefault(d, n)
                        for name, rec in ovars.items():
                            v.setdefault(name, rec)
                        for k, val in oattrs.items():
                            if k != "day":
                                attrs.setdefault(k, val)
                except Exception as e:                          # an unreadable old file is replaced, like the reference does
                    print(f"[Phyto] plankton.nc: could not carry the existing variables through ({e}).")
            if day_value is not None:
                attrs["day"] = float(day_value)
            ncio.write_nc(path, dims, v, attrs)
            return True
        except Exception as e:                                  # the reference logs and carries on
            print(f"[Phyto] save_distribution_nc failed: {e}")
            return False

    def load_distribution_nc(self, path):
        from . import ncio
        try:
            v, _ = ncio.read_nc(path, ["C_phyto_s"])
            C = np.asarray(v["C_phyto_s"], dtype=np.float64)
            if C.shape != (self.S, self.grid.n_lat, self.grid.n_lon):
                print(f"[Phyto] plankton.nc shape {C.shape} does not match (S={self.S}, grid); keeping the current state.")
                return False
            self.C_phyto_s = C
            return True
        except Exception as e:
            print(f"[Phyto] load_distribution_nc failed: {e}")
            return False


def advect_diffuse(dev, C_s, uo, vo, dt_seconds, land_mask, K_h=None, adv_alpha=None):
    """dev: qingdai_amd.device.Device (or `grid._ops()`); C_s: [S, n_lat, n_lon]; returns the new array."""
    C_s = np.array(C_s, dtype=np.float64, copy=True)
    if dt_seconds <= 0.0:
        return C_s
    K_h = float(os.getenv("QD_PHYTO_KH", os.getenv("QD_KH_OCEAN", "5.0e3"))) if K_h is None else float(K_h)
    adv_alpha = float(os.getenv("QD_PHYTO_ADV_ALPHA", "0.7")) if adv_alpha is None else float(adv_alpha)
    ocean = (np.asarray(land_mask) == 0)
    for s in range(C_s.shape[0]):
        C = C_s[s]
        C_adv = dev.op_advect(C, uo, vo, float(dt_seconds), ocean=True)
        C_new = (1.0 - adv_alpha) * C + adv_alpha * C_adv
        if K_h > 0.0:
            C_new = np.nan_to_num(C_new)
            C_new += float(dt_seconds) * K_h * dev.op_laplacian(C_new, ocean=True)
        C_new = np.clip(C_new, 0.0, np.inf)
        C_new[~ocean] = 0.0
        C_s[s] = C_new
    for j in (0, -1):
        row = ocean[j, :]
        if np.any(row):
            for s in range(C_s.shape[0]):
                C_s[s, j, row] = float(np.mean(C_s[s, j, :][row]))
    return C_s
