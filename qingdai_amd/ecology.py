"""
qingdai_amd/ecology.py -- the per-physics-step ecology on the device (stages 2-4; BASELINE config 5).

Mirrors, with the reference's names and call signatures, the part of pygcm/ecology the driver touches every step:

  EcologyAdapter(grid, land_mask).step_subdaily(I_total, cloud_eff, dt)      adapter.py:33-186
      .get_surface_albedo_bands()                                             adapter.py:519-545
  PopulationCanopy  -- the sub-daily face of PopulationManager               population.py:48-122,252-294,831-915
      E_day, LAI_layers_SK, total_LAI(), canopy_reflectance_factor(), step_subdaily()
  IndividualPool(grid, land_mask, eco).try_substep(isr_A, isr_B, eco, soil, dt, day)   individuals.py:37-191

State lives in the grid's Device (qd_eco_* / qd_indiv_* of include/qingdai_hip.h); inside a fused loop
(`Device.step_n(..., ecology=True)`) none of these methods is called at all.  The daily population dynamics
(PopulationManager.step_daily, spread, genes, IndividualPool.step_daily) are host code that runs once per planet-day and is
NOT part of this package: `Simulation` hands `E_day` to a caller-supplied daily hook and takes the new LAI layers back.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np

from . import spectral as sp
from ._lib import qd_eco_params

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


def _envf(name, default):
    try:
        return float(os.getenv(name, str(default)))
    except ValueError:
        return float(default)


def _envi(name, default):
    try:
        return int(os.getenv(name, str(default)))
    except ValueError:
        return int(default)


def _peaks_from_env(prefix):
    """`<prefix>PEAKS="450:40:0.6, 680:30:0.8"` (genes.py:53-67); the default is that two-peak absorber."""
    out = []
    for part in os.getenv(prefix + "PEAKS", "").split(","):
        try:
            c, w, h = part.strip().split(":")
            out.append((float(c), float(w), float(h)))
        except ValueError:
            continue
    return out or [(450.0, 40.0, 0.6), (680.0, 30.0, 0.8)]


class PopulationCanopy:
    """LAI layers + daily energy buffer + canopy cache of PopulationManager, resident on the device."""

    def __init__(self, dev, land_mask, diag=False):
        self._dev = dev
        self.land = (np.asarray(land_mask) == 1)
        self.shape = self.land.shape
        self.K = max(1, _envi("QD_ECO_COHORT_K", 1))
        w_env = os.getenv("QD_ECO_SPECIES_WEIGHTS", "").strip()
        ns_default = max(1, _envi("QD_ECO_NS", 20))
        if w_env:
            try:
                w = [float(x) for x in w_env.split(",") if x.strip() != ""]
            except ValueError:
                w = [1.0]
        else:
            w = [1.0 / float(ns_default)] * ns_default
        tot = sum(w) if w else 1.0
        if tot <= 0:
            self.species_weights = np.full((ns_default,), 1.0 / float(ns_default))
        else:
            self.species_weights = np.asarray([max(0.0, x) for x in w], dtype=float)
            self.species_weights /= tot
        self.Ns = int(self.species_weights.shape[0])
        lai0 = np.zeros(self.shape)
        lai0[self.land] = _envf("QD_ECO_LAI_INIT", 0.2)
        self.LAI_layers_SK = np.zeros((self.Ns, self.K) + self.shape)
        for s in range(self.Ns):
            for k in range(self.K):
                self.LAI_layers_SK[s, k] = float(self.species_weights[s]) * (lai0 / float(self.K))
        self._species_R_leaf = None
        self.push_layers(init=True)

    # -- LAI
    def push_layers(self, layers=None, init=False):
        """Hand the (changed) [S, K, lat, lon] stack to the device: what the daily step does once per planet-day."""
        if layers is not None:
            self.LAI_layers_SK = np.asarray(layers, dtype=np.float64)
        a = np.ascontiguousarray(self.LAI_layers_SK, dtype=np.float64)
        if a.shape[-2:] != self.shape:
            raise ValueError(f"LAI layers: expected [..., {self.shape[0]}, {self.shape[1]}], got {a.shape}")
        n = int(np.prod(a.shape[:-2]))
        self._dev._chk(self._dev.lib.qd_eco_set_lai_layers(self._dev.h, a.ctypes.data, n, 1 if init else 0), "qd_eco_set_lai_layers")

    def total_LAI(self):
        self._dev._host.pop("ECO_LAI", None)
        return self._dev.get("ECO_LAI").copy()

    # -- daily energy buffer
    @property
    def E_day(self):
        self._dev._host.pop("ECO_EDAY", None)
        return self._dev.get("ECO_EDAY").copy()

    @E_day.setter
    def E_day(self, arr):
        self._dev.upload_now("ECO_EDAY", np.broadcast_to(np.asarray(arr, dtype=np.float64), self.shape))

    # -- canopy
    def canopy_reflectance_factor(self):
        """f(LAI) on land, NaN elsewhere (population.py:831-842); needs a canopy cache (any sub-step or banded call builds it)."""
        self._dev._host.pop("ECO_F", None)
        out = np.full(self.shape, np.nan)
        f = self._dev.get("ECO_F")
        out[self.land] = f[self.land]
        return out

    def set_species_reflectance_bands(self, R):
        R = np.asarray(R, dtype=float)
        self._species_R_leaf = np.clip(R, 0.0, 1.0) if R.ndim == 2 else None

    def effective_leaf_reflectance_bands(self, nb):
        """R_eff[b] = clip(sum_i w_i R_i[b]) with the reference's fallbacks (population.py:856-873)."""
        R = self._species_R_leaf
        if R is None:
            return np.full((nb,), 0.5)
        if R.shape[1] != nb:
            return np.full((nb,), float(np.nanmean(R)))
        w = self.species_weights if self.species_weights.size == R.shape[0] else np.full((R.shape[0],), 1.0 / max(1, R.shape[0]))
        return np.clip(np.tensordot(w, R, axes=(0, 0)), 0.0, 1.0)

    def state(self):
        out = (ctypes.c_double * 5)()
        self._dev._chk(self._dev.lib.qd_eco_get_state(self._dev.h, out), "qd_eco_get_state")
        return {"hours": out[0], "next_recompute_hours": out[1], "step_count": int(out[2]), "n_recompute": int(out[3]),
                "alpha_cached": bool(out[4])}

    def summary(self):
        L = self.total_LAI()[self.land]
        if L.size == 0:
            return {"LAI_min": 0.0, "LAI_mean": 0.0, "LAI_max": 0.0}
        return {"LAI_min": float(np.min(L)), "LAI_mean": float(np.mean(L)), "LAI_max": float(np.max(L))}


@dataclass
class AdapterConfig:
    substep_every_nphys: int = 1
    lai_albedo_weight: float = 1.0


class EcologyAdapter:
    def __init__(self, grid, land_mask, dev=None, albedo_couple=None, f32_maps=None):
        """f32_maps: None = QD_ECO_F32 (0) -- store the canopy maps (LAI_tot, snapshot, f, alpha, banded alpha) as f32 on the device
        (BASELINE configs[4] "f32 mixed precision"); arithmetic, reductions and E_day stay f64."""
        self.grid = grid
        self._dev = dev if dev is not None else getattr(grid, "_device", None)
        if self._dev is None:
            raise RuntimeError("EcologyAdapter needs the grid's Device (create the SpectralModel first, or pass dev=)")
        self.land_mask = (np.asarray(land_mask) == 1)
        self.cfg = AdapterConfig(_envi("QD_ECO_SUBSTEP_EVERY_NPHYS", 1), _envf("QD_ECO_LAI_ALBEDO_WEIGHT", 1.0))
        self.bands = sp.make_bands()
        self.w_b = sp.band_weights_from_mode(self.bands)
        self.R_leaf = sp.default_leaf_reflectance(self.bands)
        self.alpha_leaf_scalar = float(np.sum(self.R_leaf * self.w_b))
        if albedo_couple is None:
            albedo_couple = _envi("QD_ECO_SUBDAILY_ENABLE", 1) == 1 and _envi("QD_ECO_ALBEDO_COUPLE", 1) == 1
        self.params = qd_eco_params(
            k_canopy=_envf("QD_ECO_LAI_K", 0.5), leaf_scalar=float(np.clip(self.alpha_leaf_scalar, 0.0, 1.0)),
            soil_ref=_envf("QD_ECO_SOIL_REFLECT", 0.20), w_lai=self.cfg.lai_albedo_weight,
            light_update_hours=_envf("QD_ECO_LIGHT_UPDATE_EVERY_HOURS", 6.0),
            recompute_lai_delta=_envf("QD_ECO_LIGHT_RECOMPUTE_LAI_DELTA", 0.05),
            substep_every_nphys=max(1, self.cfg.substep_every_nphys), albedo_couple=1 if albedo_couple else 0,
            bands_couple=1 if _envi("QD_ECO_BANDS_COUPLE", 0) == 1 else 0,
            water_couple=1 if (_envi("QD_PHYTO_ENABLE", 0) == 1 and _envi("QD_PHYTO_ALBEDO_COUPLE", 1) == 1) else 0,
            use_lai=1 if _envi("QD_ECO_USE_LAI", 1) == 1 else 0,
            map_f32=1 if ((_envi("QD_ECO_F32", 0) == 1) if f32_maps is None else bool(f32_maps)) else 0)
        self.configure()
        self._count = 0
        self.pop = None
        if not self.params.use_lai:        # M1 branch (adapter.py:79-80,162-166): no population, scalar leaf alpha on land
            self.species_drought_tolerance = None
            return
        self.pop = PopulationCanopy(self._dev, self.land_mask.astype(int))
        # per-species leaf reflectance and drought tolerance from the QD_ECO_SPECIES_{i}_* genes (adapter.py:90-116, genes.py:45-90)
        R, tol = species_tables(self.bands, self.pop.Ns)
        self.pop.set_species_reflectance_bands(R)
        self.species_drought_tolerance = tol

    def configure(self):
        self._dev._chk(self._dev.lib.qd_eco_configure(self._dev.h, ctypes.byref(self.params), ctypes.sizeof(self.params)),
                       "qd_eco_configure")

    def step_subdaily(self, I_total=None, cloud_eff=None, dt_seconds=300.0):
        """adapter.py:140-186.  With I_total=None the resident ISR is used (the usual case); returns the land-only alpha map
        on a sub-step boundary (downloaded), None otherwise."""
        d = self._dev
        if I_total is not None:
            d.set("ISR", I_total)
        d.flush()
        d._chk(d.lib.qd_eco_substep(d.h, float(dt_seconds)), "qd_eco_substep")
        out = (ctypes.c_double * 5)()
        d._chk(d.lib.qd_eco_get_state(d.h, out), "qd_eco_get_state")
        if int(out[2]) % max(1, self.cfg.substep_every_nphys) != 0:
            return None
        d._host.pop("ECO_ALPHA", None)
        return d.get("ECO_ALPHA").copy()

    def banded_alpha(self):
        """The driver's daily reduction (run_simulation.py:1839-1844) computed on the device and left resident in
        ECO_ALPHA_BANDED: clip(nansum_b A_b w_b, 0, 1).  Returns the host copy."""
        d = self._dev
        nb = int(self.bands.nbands)
        r = np.ascontiguousarray(self.pop.effective_leaf_reflectance_bands(nb), dtype=np.float64)
        w = np.ascontiguousarray(self.w_b, dtype=np.float64)
        d.flush()
        d._chk(d.lib.qd_eco_banded_alpha(d.h, nb, r.ctypes.data_as(_dp), w.ctypes.data_as(_dp)), "qd_eco_banded_alpha")
        d._host.pop("ECO_ALPHA_BANDED", None)
        return d.get("ECO_ALPHA_BANDED").copy()

    def get_surface_albedo_bands(self):
        """(A_b [NB, lat, lon], w_b) like adapter.py:519-545, rebuilt on the host from the resident canopy factor."""
        nb = int(self.bands.nbands)
        R_eff = self.pop.effective_leaf_reflectance_bands(nb)
        f = self.pop.canopy_reflectance_factor()
        A = np.full((nb,) + self.pop.shape, np.nan)
        land = self.land_mask
        for b in range(nb):
            A[b][land] = np.clip(R_eff[b] * f[land] + (1.0 - f[land]) * self.params.soil_ref, 0.0, 1.0)
        return A, self.w_b.copy()


def sample_pool(land_mask, species_weights, R_species, drought_tol, nb, sample_frac, per_cell):
    """The arrays IndividualPool.__init__ draws (individuals.py:63-131), as a pure host function: sampled cells without
    replacement from default_rng(42), the cell index of every individual, species by weight, per-band coefficients = species
    leaf reflectance + N(0, 0.02) jitter clipped to [0, 1], drought tolerance by species.  Same generator calls in the same
    order as the reference, so the same land mask and species table give the same pool."""
    land = (np.asarray(land_mask) == 1)
    width = land.shape[1]
    w = np.asarray(species_weights, dtype=float)
    if w.ndim != 1 or w.size <= 0:
        w = np.asarray([1.0])
    ns = int(w.size)
    spw = w / w.sum() if w.sum() > 0 else np.full((ns,), 1.0 / ns)
    rng = np.random.default_rng(seed=42)
    land_idx = np.flatnonzero(land.ravel())
    want = max(1, int(sample_frac * land_idx.size))
    picked = land_idx if want >= land_idx.size else rng.choice(land_idx, size=want, replace=False)
    sample_j = np.asarray(picked // width, dtype=np.int32)
    sample_i = np.asarray(picked % width, dtype=np.int32)
    n_cells = int(sample_j.size)
    n_indiv = n_cells * int(per_cell)
    cell = np.repeat(np.arange(n_cells, dtype=np.int32), int(per_cell))
    species_id = rng.choice(np.arange(ns, dtype=np.int32), size=n_indiv, p=spw)
    R = R_species
    if R is None or R.shape[0] != ns:
        R = np.full((ns, nb), 0.5)
    if R.shape[1] > nb:
        R = R[:, :nb]
    elif R.shape[1] < nb:
        R = np.pad(R, ((0, 0), (0, nb - R.shape[1])), mode="edge")
    Ab = np.clip(R[species_id, :] + rng.normal(0.0, 0.02, size=(n_indiv, nb)), 0.0, 1.0)
    tol = np.full((ns,), 0.5) if drought_tol is None or len(drought_tol) != ns else np.asarray(drought_tol, dtype=float)
    return {"sp_weights": spw, "sample_j": sample_j, "sample_i": sample_i, "indiv_cell_index": cell, "indiv_species_id": species_id,
            "indiv_Ab": Ab, "indiv_tol": np.clip(tol, 0.0, 1.0)[species_id]}


def species_tables(bands, n_species):
    """Per-species leaf reflectance [Ns, NB] and drought tolerance [Ns] from the QD_ECO_SPECIES_{i}_* genes
    (adapter.py:90-116, genes.py:45-111)."""
    R, tol = [], []
    for i in range(n_species):
        pre = f"QD_ECO_SPECIES_{i}_"
        R.append(np.clip(1.0 - sp.absorbance_from_peaks(bands, _peaks_from_env(pre)), 0.0, 1.0))
        tol.append(_envf(pre + "DROUGHT_TOL", 0.3))
    return np.stack(R, axis=0), np.asarray(tol, dtype=float)


class IndividualPool:
    """Sampled individuals (individuals.py:37-191).  The sampling uses the same generator calls in the same order as the
    reference (default_rng(42): cells without replacement, species by weight, N(0, 0.02) jitter), so a given land mask yields
    the same pool."""

    def __init__(self, grid, land_mask, eco_adapter, *, sample_frac=0.02, per_cell=100, substeps_per_day=10, day_seconds=None,
                 soil_cap=None, diag=False, f32_storage=None):
        if getattr(eco_adapter, "pop", None) is None:
            raise RuntimeError("IndividualPool requires EcologyAdapter.pop (QD_ECO_USE_LAI=1)")     # individuals.py:67-69
        self._dev = eco_adapter._dev
        self.land_mask = (np.asarray(land_mask) == 1)
        self.h, self.w = self.land_mask.shape
        self.bands = eco_adapter.bands
        self.nb = int(self.bands.nbands)
        pop = eco_adapter.pop
        frac = _envf("QD_ECO_INDIV_SAMPLE_FRAC", sample_frac)
        self.per_cell = _envi("QD_ECO_INDIV_PER_CELL", per_cell)
        self.substeps_per_day = max(1, _envi("QD_ECO_INDIV_SUBSTEPS_PER_DAY", substeps_per_day))
        arr = sample_pool(self.land_mask, pop.species_weights, pop._species_R_leaf,
                          getattr(eco_adapter, "species_drought_tolerance", None), self.nb, frac, self.per_cell)
        for k, v in arr.items():
            setattr(self, k, v)
        self.ns = int(self.sp_weights.size)
        self.n_cells = int(self.sample_j.size)
        self.n_indiv = self.n_cells * self.per_cell
        from .forcing import PLANET_OMEGA
        self.day_seconds = float(day_seconds) if day_seconds else 2 * np.pi / PLANET_OMEGA
        self.soil_cap = float(soil_cap) if soil_cap is not None else _envf("QD_ECO_SOIL_WATER_CAP", 50.0)
        # QD_ECO_F32=1: keep the [N, NB] coefficient table as f32 on the device (BASELINE configs[4] "f32 mixed precision")
        self.f32_storage = (_envi("QD_ECO_F32", 0) == 1) if f32_storage is None else bool(f32_storage)
        self.configure()

    def configure(self, **star_kw):
        d = self._dev
        specA, specB, tray = (np.ascontiguousarray(a, dtype=np.float64) for a in sp.star_band_weights(self.bands, **star_kw))
        sj, si, ci = (np.ascontiguousarray(a, dtype=np.int32) for a in (self.sample_j, self.sample_i, self.indiv_cell_index))
        Ab = np.ascontiguousarray(self.indiv_Ab, dtype=np.float64)
        tol = np.ascontiguousarray(self.indiv_tol, dtype=np.float64)
        d._chk(d.lib.qd_indiv_configure(d.h, self.n_cells, sj.ctypes.data_as(_ip), si.ctypes.data_as(_ip), self.n_indiv,
                                        ci.ctypes.data_as(_ip), Ab.ctypes.data, tol.ctypes.data, self.nb, specA.ctypes.data_as(_dp),
                                        specB.ctypes.data_as(_dp), tray.ctypes.data_as(_dp), self.substeps_per_day,
                                        self.day_seconds, self.soil_cap, 1 if self.f32_storage else 0), "qd_indiv_configure")

    def try_substep(self, isr_A=None, isr_B=None, eco_adapter=None, soil_W_land=None, dt_seconds=300.0, day_length_seconds=None):
        """individuals.py:142-191 on the resident ISR_A / ISR_B / W_LAND (arrays given here are uploaded first; `soil_W_land`
        is the soil INDEX the reference driver passes, so it is uploaded scaled back by the cap).  Returns True when a
        sub-step fired."""
        d = self._dev
        if isr_A is not None:
            d.set("ISR_A", isr_A)
        if isr_B is not None:
            d.set("ISR_B", isr_B)
        if soil_W_land is not None:
            d.set("W_LAND", np.asarray(soil_W_land, dtype=np.float64) * max(1e-6, self.soil_cap))
        d.flush()
        fired = ctypes.c_int32(0)
        d._chk(d.lib.qd_indiv_substep(d.h, float(dt_seconds), ctypes.byref(fired)), "qd_indiv_substep")
        return bool(fired.value)

    def _pull(self):
        E, S = np.empty(self.n_indiv), np.empty(self.n_indiv)
        self._dev._chk(self._dev.lib.qd_indiv_download(self._dev.h, E.ctypes.data, S.ctypes.data), "qd_indiv_download")
        return E, S

    @property
    def indiv_E_day(self):
        return self._pull()[0]

    @property
    def indiv_water_stress_days(self):
        return self._pull()[1]

    def reset(self, E_day=None, stress_days=None):
        """What the daily step does to the buffers (individuals.py:207-208 and the end of step_daily)."""
        E = np.zeros(self.n_indiv) if E_day is None else np.ascontiguousarray(E_day, dtype=np.float64)
        S = np.zeros(self.n_indiv) if stress_days is None else np.ascontiguousarray(stress_days, dtype=np.float64)
        self._dev._chk(self._dev.lib.qd_indiv_upload(self._dev.h, E.ctypes.data, S.ctypes.data), "qd_indiv_upload")
