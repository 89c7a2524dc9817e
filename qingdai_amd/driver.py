"""
qingdai_amd/driver.py -- drop-in for the reference driver `scripts/run_simulation.py:main()`
(run_simulation.py:1161-2523) with the per-timestep loop resident on the MI355X.

Kept from the reference driver: the QD_* environment surface of the path (plus QD_N_LAT / QD_N_LON,
which the reference only honours in its facade and topography generator -- SURVEY.md section 0.1),
procedural seed-42 or NetCDF topography, slab-ocean heat capacities, restart load/save
(QD_RESTART_IN / QD_RESTART_OUT, variables u v h T_s cloud_cover q h_ice uo vo eta Ts W_land S_snow
C_snow land_mask as f4 + scalar t_seconds, run_simulation.py:63-183), SIGINT/SIGTERM/atexit autosave
(run_simulation.py:1689-1706, exit codes 130/143), QD_SIM_DAYS / QD_TOTAL_YEARS / QD_DT_SECONDS,
QD_USE_OCEAN, the QD_USE_OO(_STRICT) short-circuit, periodic diagnostics.
Ecology (QD_ECO_ENABLE, default on like the reference): the per-step part -- EcologyAdapter.step_subdaily, the alpha blend
into the base albedo, IndividualPool.try_substep -- runs inside the resident loop (qingdai_amd/ecology.py, qd_eco_*); the
DAILY population dynamics are host code outside this package, reached through `Simulation(daily_hook=...)`, which is called
where the reference calls eco.step_daily (run_simulation.py:1786-1864).  Without a hook the LAI stays at its initial value.
Phytoplankton (QD_PHYTO_ENABLE and QD_PHYTO_ADVECTION, default on like the reference, needs the ocean): the per-step transport of
the tracers by the ocean currents (phyto.advect_diffuse, run_simulation.py:2254-2258) runs inside the resident loop on resident
tracers (qingdai_amd/phyto.py PhytoTracers, qd_phyto_*); data/plankton.nc carries C_phyto_s through autosave / startup load.
Not carried over (out of the hot path): phytoplankton daily growth / optics, river routing, genes /
diversity, matplotlib panels (a note is printed instead of a plot).

Per iteration (run_simulation.py:1760-2340), all on the device through one qd_step_n call per chunk:
  hybrid precipitation -> clouds -> cloud tracer -> insolation -> P019 lapse/snow -> albedo -> Teq ->
  SpectralModel.time_step(Teq, dt) [no albedo argument, like the reference driver] -> ocean coupling ->
  snow commit + land bucket.
"""
from __future__ import annotations

import atexit
import os
import signal
import sys
import time

import numpy as np

from . import SphericalGrid, SpectralModel, WindDrivenSlabOcean, OrbitalSystem, ThermalForcing, QdParams
from . import topography as topo

PLANET_OMEGA = 8.726646259971648e-5

RESTART_VARS = {  # restart name -> device field (run_simulation.py:63-124)
    "u": "U", "v": "V", "h": "H", "T_s": "TS", "cloud_cover": "CLOUD", "q": "Q", "h_ice": "HICE",
    "uo": "UO", "vo": "VO", "eta": "ETA", "Ts": "SST", "W_land": "W_LAND", "S_snow": "S_SNOW", "C_snow": "C_SNOW",
}


# ------------------------------------------------------------------------------------- NetCDF I/O
from . import ncio  # noqa: E402


def _nc_backend():
    return ncio.backend()


OCEAN_VARS = ("uo", "vo", "eta", "Ts")


def save_restart(path, grid, dev, t_seconds, land_mask, with_ocean=True):
    """run_simulation.py:63-124: dims lat/lon, every state variable AND land_mask as f4 (the reference's `wvar` writes them all
    through one f4 helper, :86-111), the ocean variables only when an ocean exists (:99-103), scalar t_seconds (f8), format=v1.
    netCDF4 when importable, else NetCDF-3 through scipy.io."""
    v = {"lat": ("f4", ("lat",), np.asarray(grid.lat, np.float32)), "lon": ("f4", ("lon",), np.asarray(grid.lon, np.float32))}
    for name, fid in RESTART_VARS.items():
        if name in OCEAN_VARS and not with_ocean:
            continue
        v[name] = ("f4", ("lat", "lon"), dev.get(fid).astype(np.float32))
    v["land_mask"] = ("f4", ("lat", "lon"), np.asarray(land_mask, np.float32))
    v["t_seconds"] = ("f8", (), float(t_seconds))
    ncio.write_nc(path, {"lat": grid.n_lat, "lon": grid.n_lon}, v,
                  {"title": "Qingdai GCM Restart", "creator": "qingdai_amd", "format": "v1"})


def load_restart(path):
    """run_simulation.py:161-183: returns {name: float32 array} + t_seconds (arrays come back f4)."""
    v, _ = ncio.read_nc(path, list(RESTART_VARS) + ["land_mask", "t_seconds"])
    out = {k: a for k, a in v.items() if k != "t_seconds"}
    out["t_seconds"] = float(v["t_seconds"]) if "t_seconds" in v else 0.0
    return out


def save_ocean(path, grid, dev, day_value=None):
    """data/ocean.nc (run_simulation.py:185-220): uo, vo, eta, Ts as f4, attribute `day`.  Never raises."""
    try:
        v = {"lat": ("f4", ("lat",), np.asarray(grid.lat, np.float32)), "lon": ("f4", ("lon",), np.asarray(grid.lon, np.float32))}
        for name, fid in (("uo", "UO"), ("vo", "VO"), ("eta", "ETA"), ("Ts", "SST")):
            v[name] = ("f4", ("lat", "lon"), dev.get(fid).astype(np.float32))
        attrs = {"title": "Qingdai Ocean State", "source": "qingdai_amd"}
        if day_value is not None:
            attrs["day"] = float(day_value)
        ncio.write_nc(path, {"lat": grid.n_lat, "lon": grid.n_lon}, v, attrs)
        return True
    except Exception as e:                                  # the reference logs and carries on
        print(f"[Ocean] Save failed: {e}")
        return False


def load_ocean(path):
    """run_simulation.py:222-246: {uo, vo, eta, Ts, day}; missing entries are None.  Never raises."""
    out = {"uo": None, "vo": None, "eta": None, "Ts": None, "day": None}
    try:
        v, attrs = ncio.read_nc(path, ["uo", "vo", "eta", "Ts"])
        out.update({k: v.get(k) for k in ("uo", "vo", "eta", "Ts")})
        out["day"] = float(attrs["day"]) if "day" in attrs else None
    except Exception as e:
        print(f"[Ocean] Load failed '{path}': {e}")
    return out


# ------------------------------------------------------------------------------------- simulation
class Simulation:
    """The reference driver's state + loop, device resident."""

    def __init__(self, n_lat=None, n_lon=None, params: QdParams | None = None, use_ocean=None, quiet=False, device=0,
                 ecology=None, individuals=None, daily_hook=None, phyto=None):
        """ecology / individuals: None = QD_ECO_ENABLE / QD_ECO_INDIV_ENABLE (both default 1, run_simulation.py:1324,1404).
        daily_hook(sim, soil_idx, glacier_mask): the host-side daily ecology, called at planet-day boundaries.
        phyto: None = QD_PHYTO_ENABLE and QD_PHYTO_ADVECTION (both default 1, run_simulation.py:1347,1351)."""
        env = os.environ
        n_lat = int(n_lat if n_lat is not None else env.get("QD_N_LAT", "121"))   # run_simulation.py:1195 is 121x240
        n_lon = int(n_lon if n_lon is not None else env.get("QD_N_LON", "240"))
        self.quiet = quiet
        self.grid = SphericalGrid(n_lat, n_lon)
        # run_simulation.py:1198-1215: external topography (QD_TOPO_NC) or the procedural seed-42 planet, for
        # which the reference driver has NO elevation map (orography / lapse then see a flat bed)
        topo_nc = env.get("QD_TOPO_NC")
        self.elevation = None
        loaded = False
        if topo_nc and os.path.exists(topo_nc):
            try:
                self.elevation, self.land_mask, self.base_albedo, self.friction = topo.load_topography_from_netcdf(
                    topo_nc, self.grid, quiet=quiet)
                loaded = True
            except Exception as e:
                print(f"[Topo] Failed to load '{topo_nc}': {e}\nFalling back to procedural generation.")
        if not loaded:
            self.land_mask = topo.create_land_sea_mask(self.grid)
            self.base_albedo, self.friction = topo.generate_base_properties(self.land_mask)
            if not quiet:
                w = np.cos(np.deg2rad(self.grid.lat_mesh))
                frac = float((w * (self.land_mask == 1)).sum() / (w.sum() + 1e-15))
                print(f"[Topo] Procedural topography (seed 42). Land fraction: {frac:.3f}")
        rho_w = float(env.get("QD_RHO_W", "1000"))
        cp_w = float(env.get("QD_CP_W", "4200"))
        H_mld = float(env.get("QD_MLD_M", "50"))
        Cs_ocean = rho_w * cp_w * H_mld
        Cs_land = float(env.get("QD_CS_LAND", "3e6"))
        Cs_ice = float(env.get("QD_CS_ICE", "5e6"))
        p = params or QdParams.from_env()
        self.gcm = SpectralModel(self.grid, self.friction, H=8000, tau_rad=10 * 24 * 3600,
                                 greenhouse_factor=float(env.get("QD_GH_FACTOR", "0.40")),
                                 C_s_map=np.where(self.land_mask == 1, Cs_land, Cs_ocean).astype(float),
                                 land_mask=self.land_mask, Cs_ocean=Cs_ocean, Cs_land=Cs_land, Cs_ice=Cs_ice,
                                 params=p, device=device)
        self.dev = self.gcm._dev
        self.dev.upload_now("BASE_ALBEDO", self.base_albedo)
        if self.elevation is not None:
            self.dev.upload_now("ELEVATION", np.nan_to_num(self.elevation))
        use_ocean = (int(env.get("QD_USE_OCEAN", "1")) == 1) if use_ocean is None else bool(use_ocean)
        self.ocean = None
        if use_ocean:
            H_ocean = float(env.get("QD_OCEAN_H_M", str(H_mld)))
            self.ocean = WindDrivenSlabOcean(self.grid, self.land_mask, H_ocean,
                                             init_Ts=np.where(self.land_mask == 0, 288.0, 288.0))
        self.forcing = ThermalForcing(self.grid, OrbitalSystem())
        self.t = 0.0
        self._step_index = 0
        self.dt = int(env.get("QD_DT_SECONDS", "300"))
        # ecology (run_simulation.py:1324-1423): adapter + canopy population + sampled individuals, state on the device
        self.eco = self.indiv = None
        self.daily_hook = daily_hook
        self.day_seconds = 2 * np.pi / PLANET_OMEGA
        self._accum_day = 0.0
        eco_on = (int(env.get("QD_ECO_ENABLE", "1")) == 1) if ecology is None else bool(ecology)
        if eco_on:
            from .ecology import EcologyAdapter, IndividualPool
            self.eco = EcologyAdapter(self.grid, self.land_mask, dev=self.dev)
            ind_on = (int(env.get("QD_ECO_INDIV_ENABLE", "1")) == 1) if individuals is None else bool(individuals)
            if ind_on and self.eco.pop is not None:
                self.indiv = IndividualPool(self.grid, self.land_mask, self.eco, sample_frac=0.02, per_cell=150,
                                            substeps_per_day=10, day_seconds=self.day_seconds)
            if not quiet:
                lai = f"LAI mean {self.eco.pop.summary()['LAI_mean']:.2f}" if self.eco.pop is not None else "no population (M1)"
                print(f"[Ecology] device sub-step: NB={self.eco.bands.nbands}, alpha_leaf={self.eco.alpha_leaf_scalar:.3f}, "
                      f"{lai}, individuals {self.indiv.n_indiv if self.indiv else 0}")
        # phytoplankton tracers (run_simulation.py:1346-1364): only their transport by the currents is on this path
        self.phyto = None
        if phyto is None:
            phyto = int(env.get("QD_PHYTO_ENABLE", "1")) == 1 and int(env.get("QD_PHYTO_ADVECTION", "1")) == 1
        if phyto and self.ocean is not None:
            from .phyto import PhytoTracers
            self.phyto = PhytoTracers(self.grid, self.land_mask, dev=self.dev)
            if not quiet:
                print(f"[Phyto] resident tracers: S={self.phyto.S}, K_h={self.phyto.K_h:g} m^2/s, adv_alpha={self.phyto.adv_alpha:g}")
        # banded initial surface temperature (run_simulation.py:310-328)
        if int(env.get("QD_INIT_BANDED", "0")) == 1:
            T_eq, T_pole = float(env.get("QD_INIT_T_EQ", "295.0")), float(env.get("QD_INIT_T_POLE", "265.0"))
            Ts0 = T_pole + (T_eq - T_pole) * (np.cos(np.deg2rad(self.grid.lat_mesh)) ** 2)
            self.gcm.T_s = Ts0.copy()
            if self.ocean is not None:
                self.ocean.Ts = np.where(self.land_mask == 0, Ts0, 288.0)

    # -- restart
    def load(self, path):
        rst = load_restart(path)
        for name, fid in RESTART_VARS.items():
            if name in rst and (self.ocean is not None or name not in ("uo", "vo", "eta", "Ts")):
                arr = np.asarray(rst[name], dtype=np.float64)        # arrives as f4, like the reference
                if name == "cloud_cover":
                    arr = np.clip(arr, 0.0, 1.0)
                if name == "h_ice":
                    arr = np.maximum(arr, 0.0)
                self.dev.set(fid, arr)
        self.t = float(rst.get("t_seconds", 0.0))

    def save(self, path):
        save_restart(path, self.grid, self.dev, self.t, self.land_mask, with_ocean=self.ocean is not None)

    def load_ocean_override(self, path):
        """run_simulation.py:1497-1509 / 1542-1553: QD_LOAD_OCEAN=1 (default) lets a standardized data/ocean.nc override the ocean
        fields after the atmosphere checkpoint was read.  Returns True when something was loaded."""
        if self.ocean is None or not os.path.exists(path):
            return False
        v, _ = ncio.read_nc(path, list(OCEAN_VARS))
        for name in OCEAN_VARS:
            if name in v:
                self.dev.set(RESTART_VARS[name], np.asarray(v[name], dtype=np.float64))
        return bool(v)

    def save_autosave(self, data_dir="data"):
        """run_simulation.py:248-270 + 126-159 + 185-220: data/atmosphere.nc (the restart layout with the epoch
        in t_seconds), data/ocean.nc, data/topography.nc.  Ecology / genes files are outside this path."""
        day = self.t / (2 * np.pi / PLANET_OMEGA)
        save_restart(os.path.join(data_dir, "atmosphere.nc"), self.grid, self.dev, self.t, self.land_mask, with_ocean=self.ocean is not None)
        if self.ocean is not None:
            save_ocean(os.path.join(data_dir, "ocean.nc"), self.grid, self.dev, day_value=day)
        topo.export_topography_to_netcdf(os.path.join(data_dir, "topography.nc"), self.grid, self.land_mask, self.base_albedo,
                                         self.friction, elevation=self.elevation)
        if self.phyto is not None:                              # run_simulation.py:1677-1685 (the tracer part of plankton.nc)
            self.phyto.save_distribution_nc(os.path.join(data_dir, "plankton.nc"), day_value=day)

    # -- the loop
    def bootstrap_ecology(self):
        """run_simulation.py:1716-1726: one step_subdaily on the t = 0 insolation before the loop starts."""
        if self.eco is None or not self.eco.params.albedo_couple:
            return
        self.forcing.update_device(0.0, with_teq=False)
        self.eco.step_subdaily(None, None, float(self.dt))

    def _daily(self):
        """The day-boundary block of run_simulation.py:1786-1864 for the ecology: soil index from W_land, zero on ice sheets."""
        if self.daily_hook is None:
            return
        cap = float(os.environ.get("QD_ECO_SOIL_WATER_CAP", "50.0"))
        for k in ("W_LAND", "GLACIER"):
            self.dev._host.pop(k, None)
        glacier = self.dev.get("GLACIER") != 0.0
        soil_idx = np.clip(self.dev.get("W_LAND") / max(1e-6, cap), 0.0, 1.0) * (~glacier)
        self.daily_hook(self, soil_idx, glacier)

    def run_steps(self, n):
        """n iterations of run_simulation.py:1760-2340 as resident qd_step_n calls: one per stretch between planet-day
        boundaries when a daily ecology hook is installed (the hook runs where the reference runs eco.step_daily, i.e. before
        the rest of the step that completes the day), one per QD_ENERGY_TUNE_EVERY steps when the autotuner is on."""
        if n <= 0:
            return
        if self.eco is not None and self.daily_hook is not None:
            left = n
            while left > 0:
                a, k = self._accum_day, 0                      # run_simulation.py:1784-1786: accum += dt; while accum >= day
                while k < left and a + self.dt < self.day_seconds:
                    a += self.dt
                    k += 1
                if k > 0:
                    self._run_span(k)
                    self._accum_day = a
                    left -= k
                    continue
                self._accum_day += self.dt
                while self._accum_day >= self.day_seconds:
                    self._accum_day -= self.day_seconds
                    self._daily()
                self._run_span(1)
                left -= 1
            return
        self._run_span(n)

    def _run_span(self, n):
        p = self.dev.params
        autotune = (int(p.gh_lock) == 0) and int(os.environ.get("QD_ENERGY_AUTOTUNE", "0")) == 1 and self.ocean is not None
        if autotune:                                           # run_simulation.py:1256-1257, 2242-2246
            # the reference evaluates the budget inside step i (i % every == 0), after time_step, and the nudged
            # parameters first act on step i+1: run that step on its own with the in-step diagnostics, then the rest
            from . import energy as _energy
            every = max(1, int(os.environ.get("QD_ENERGY_TUNE_EVERY", "50")))
            left = n
            while left > 0:
                if self._step_index % every == 0:
                    self._run_chunk(1, energy_diag=True)
                    _energy.autotune_greenhouse_params(p, self.dev.energy_diagnostics_last())
                    self.dev.push_params()
                    left -= 1
                    continue
                k = min(left, every - (self._step_index % every))
                self._run_chunk(k)
                left -= k
            return
        self._run_chunk(n)

    def _run_chunk(self, n, energy_diag=False):
        times = self.t + self.dt * np.arange(n)
        stars = self.forcing.star_table(times)
        self.dev.step_n(stars, float(self.dt), with_ocean=self.ocean is not None, with_physics=True, pass_albedo=False,
                        with_hydrology=True, energy_diag=energy_diag, ecology=self.eco is not None, phyto=self.phyto is not None)
        self.t = float(times[-1] + self.dt)
        self._step_index += n

    def diagnostics(self):
        d = self.dev
        from ._lib import R_MAXABS, R_COSMEAN
        return {"max|u|": d.reduce("U", R_MAXABS), "max|v|": d.reduce("V", R_MAXABS), "max|h|": d.reduce("H", R_MAXABS),
                "<T_s>": d.reduce("TS", R_COSMEAN), "<cloud>": d.reduce("CLOUD", R_COSMEAN),
                "<E>": d.reduce("EFLUX", R_COSMEAN), "<P>": d.reduce("PRECIP", R_COSMEAN)}


def chunk_until(t, dt, next_autosave_t, remaining, max_chunk=200):
    """Steps to hand to the device loop in one go: at most `max_chunk`, at most `remaining`, and -- when a periodic autosave is
    pending -- exactly up to the step whose END reaches the threshold (the reference tests `t >= next_autosave_t` at the top of the
    following step, run_simulation.py:1762), at least one."""
    n = min(max_chunk, remaining)
    if next_autosave_t is not None:
        to_thr = int(np.ceil((next_autosave_t - t) / dt - 1e-9))
        if to_thr > 0:
            n = max(1, min(n, to_thr))
    return n


def main(argv=None):
    env = os.environ
    print("--- Initializing Qingdai GCM (MI355X device path) ---")
    # P020 Phase-0 switch (run_simulation.py:1172-1191): the facade only advances a clock
    if int(env.get("QD_USE_OO", "0")) == 1 and int(env.get("QD_USE_OO_STRICT", "0")) == 1:
        print("[P020] QD_USE_OO=1 QD_USE_OO_STRICT=1 -> facade stub only; exiting legacy engine.")
        return 0
    sim = Simulation()
    day = 2 * np.pi / PLANET_OMEGA
    if env.get("QD_TOTAL_YEARS"):
        duration = float(env["QD_TOTAL_YEARS"]) * sim.forcing.orbital_system.T_planet
    elif env.get("QD_SIM_DAYS"):
        duration = float(env["QD_SIM_DAYS"]) * day
    else:
        duration = 5 * sim.forcing.orbital_system.T_planet
    # run_simulation.py:1451-1563: QD_RESTART_IN wins; else the autosave checkpoint data/atmosphere.nc when QD_AUTOSAVE_LOAD=1 (the
    # default); in both cases data/ocean.nc then overrides the ocean fields when QD_LOAD_OCEAN=1 (default).  Load failures fall back
    # to a fresh start, as in the reference.
    data_dir = env.get("QD_DATA_DIR", "data")
    restart_in = env.get("QD_RESTART_IN")
    autosave_nc = os.path.join(data_dir, "atmosphere.nc")
    loaded = None
    if restart_in and os.path.exists(restart_in):
        loaded = restart_in
    elif not restart_in and int(env.get("QD_AUTOSAVE_LOAD", "1")) == 1 and os.path.exists(autosave_nc):
        loaded = autosave_nc
    if loaded:
        try:
            sim.load(loaded)
            print(f"[{'Restart' if loaded == restart_in else 'Autosave'}] loaded '{loaded}' at t={sim.t:.1f} s")
            if int(env.get("QD_LOAD_OCEAN", "1")) == 1:
                try:
                    if sim.load_ocean_override(os.path.join(data_dir, "ocean.nc")):
                        print("[Restart] Ocean state overridden from 'data/ocean.nc'.")
                except Exception as e:     # noqa: BLE001
                    print(f"[Restart] ocean.nc load skipped: {e}")
        except Exception as e:             # noqa: BLE001
            print(f"[Restart] Failed to load '{loaded}': {e}\nContinuing with fresh init.")
            loaded = None
    # run_simulation.py:1377-1399: data/plankton.nc restores the tracer distributions at startup (QD_LOAD_PLANKTON=1, default)
    if sim.phyto is not None and int(env.get("QD_LOAD_PLANKTON", "1")) == 1:
        pnc = os.path.join(data_dir, "plankton.nc")
        if os.path.exists(pnc):
            print(f"[Phyto] plankton.nc load {'OK' if sim.phyto.load_distribution_nc(pnc) else 'skipped/failed'}.")
    if not loaded and sim.t == 0.0:
        if env.get("QD_ORBIT_EPOCH_SECONDS"):
            sim.t = float(env["QD_ORBIT_EPOCH_SECONDS"])
        elif env.get("QD_ORBIT_EPOCH_DAYS"):
            sim.t = float(env["QD_ORBIT_EPOCH_DAYS"]) * day
    sim.bootstrap_ecology()
    t0 = sim.t
    n_total = len(np.arange(t0, t0 + duration, sim.dt))
    print(f"Grid resolution: {sim.grid.n_lat} lat x {sim.grid.n_lon} lon | dt = {sim.dt} s | "
          f"{duration / day:.1f} planetary days | {n_total} steps")
    print("[Plots] matplotlib panels are not produced by the device driver (out of the hot path).")

    autosave_on = int(env.get("QD_AUTOSAVE_ENABLE", "1")) == 1
    restart_out = env.get("QD_RESTART_OUT") or os.path.join("data", "restart_autosave.nc")
    # periodic autosave (run_simulation.py:1751-1764): every QD_ECO_AUTOSAVE_EVERY_HOURS PLANETARY hours (day / 24; default 6),
    # tracked as a time threshold -- the first step whose time has reached it saves, whatever the chunking of the device loop
    try:
        every_h = float(env.get("QD_ECO_AUTOSAVE_EVERY_HOURS", "6"))
        autosave_dt = every_h * (day / 24.0) if every_h > 0 else None
    except ValueError:
        autosave_dt = None
    next_autosave_t = t0 + autosave_dt if autosave_dt else None
    state = {"saved": False}

    def _autosave(reason):
        if state["saved"] or not autosave_on:
            return
        try:
            # run_simulation.py:1669-1687: data/atmosphere.nc (+ data/ocean.nc, data/topography.nc); QD_RESTART_OUT on top
            sim.save_autosave(data_dir)
            if env.get("QD_RESTART_OUT"):
                sim.save(restart_out)
            print(f"[Autosave] ({reason}) core state saved to 'data/atmosphere.nc' at t={sim.t:.1f} s")
        except Exception as e:     # noqa: BLE001  (the reference never lets I/O kill the run)
            print(f"[Autosave] skipped: {e}")
        state["saved"] = True

    def _on_signal(signum, _frame):
        _autosave("signal")
        sys.exit(130 if signum == signal.SIGINT else 143)
    signal.signal(signal.SIGINT, _on_signal)
    signal.signal(signal.SIGTERM, _on_signal)
    atexit.register(lambda: _autosave("atexit"))

    done = 0
    wall0 = time.perf_counter()
    while done < n_total:
        n = chunk_until(sim.t, sim.dt, next_autosave_t if autosave_on else None, n_total - done)
        sim.run_steps(n)
        done += n
        if int(env.get("QD_DYN_DIAG_PRINT", "1")) == 1:
            dg = sim.diagnostics()
            el = time.perf_counter() - wall0
            print(f"t={sim.t / day:8.2f} d | " + " ".join(f"{k}={v:.4g}" for k, v in dg.items()) +
                  f" | {done / max(el, 1e-9):.0f} steps/s")
        if autosave_on and next_autosave_t is not None and sim.t >= next_autosave_t - 1e-9 * sim.dt and done < n_total:
            state["saved"] = False
            _autosave("periodic")
            state["saved"] = False
            while next_autosave_t <= sim.t + 1e-9 * sim.dt:
                next_autosave_t += autosave_dt
    if env.get("QD_RESTART_OUT"):
        sim.save(env["QD_RESTART_OUT"])
        print(f"[Restart] wrote {env['QD_RESTART_OUT']}")
    # the reference leaves the final checkpoint to its atexit handler (run_simulation.py:1669-1706); doing it here gives the
    # same files to a caller that invokes main() in-process (the handler then finds the state saved)
    state["saved"] = False
    _autosave("final")
    print("--- Simulation Finished ---")
    return 0


if __name__ == "__main__":
    sys.exit(main())
