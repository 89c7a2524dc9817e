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
Not carried over (out of the hot path): ecology, phytoplankton, river routing,
matplotlib panels (a note is printed instead of a plot).

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
def _nc_backend():
    try:
        import netCDF4  # noqa: F401
        return "netCDF4"
    except Exception:
        return "scipy"


def save_restart(path, grid, dev, t_seconds, land_mask):
    """run_simulation.py:63-124: dims lat/lon, f4 state variables, scalar t_seconds (f8), format=v1.
    netCDF4 when importable, else NetCDF-3 classic through scipy.io (which has no u1: land_mask is i1)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    fields = {name: dev.get(fid).astype(np.float32) for name, fid in RESTART_VARS.items()}
    if _nc_backend() == "netCDF4":
        from netCDF4 import Dataset
        with Dataset(path, "w") as ds:
            ds.createDimension("lat", grid.n_lat)
            ds.createDimension("lon", grid.n_lon)
            ds.createVariable("lat", "f4", ("lat",))[:] = grid.lat
            ds.createVariable("lon", "f4", ("lon",))[:] = grid.lon
            for name, arr in fields.items():
                ds.createVariable(name, "f4", ("lat", "lon"))[:] = arr
            ds.createVariable("land_mask", "u1", ("lat", "lon"))[:] = land_mask.astype(np.uint8)
            ds.createVariable("t_seconds", "f8", ())[...] = float(t_seconds)
            ds.title = "Qingdai GCM Restart"
            ds.creator = "qingdai_amd"
            ds.format = "v1"
    else:
        from scipy.io import netcdf_file
        with netcdf_file(path, "w", version=2) as ds:
            ds.createDimension("lat", grid.n_lat)
            ds.createDimension("lon", grid.n_lon)
            v = ds.createVariable("lat", "f4", ("lat",)); v[:] = grid.lat.astype(np.float32)
            v = ds.createVariable("lon", "f4", ("lon",)); v[:] = grid.lon.astype(np.float32)
            for name, arr in fields.items():
                v = ds.createVariable(name, "f4", ("lat", "lon")); v[:] = arr
            v = ds.createVariable("land_mask", "i1", ("lat", "lon")); v[:] = land_mask.astype(np.int8)
            v = ds.createVariable("t_seconds", "f8", ()); v[()] = float(t_seconds)
            ds.title = b"Qingdai GCM Restart"
            ds.creator = b"qingdai_amd"
            ds.format = b"v1"


def load_restart(path):
    """run_simulation.py:126-183: returns {name: float32 array} + t_seconds (arrays come back f4)."""
    out = {}
    if _nc_backend() == "netCDF4":
        from netCDF4 import Dataset
        with Dataset(path, "r") as ds:
            for name in list(RESTART_VARS) + ["land_mask"]:
                if name in ds.variables:
                    out[name] = np.array(ds.variables[name][:])
            out["t_seconds"] = float(ds.variables["t_seconds"][...]) if "t_seconds" in ds.variables else 0.0
    else:
        from scipy.io import netcdf_file
        with netcdf_file(path, "r", mmap=False) as ds:
            for name in list(RESTART_VARS) + ["land_mask"]:
                if name in ds.variables:
                    a = np.array(ds.variables[name][:])
                    out[name] = a.astype(a.dtype.newbyteorder("="))       # classic NetCDF is big-endian on disk
            out["t_seconds"] = float(ds.variables["t_seconds"].getValue()) if "t_seconds" in ds.variables else 0.0
    return out


# ------------------------------------------------------------------------------------- simulation
class Simulation:
    """The reference driver's state + loop, device resident."""

    def __init__(self, n_lat=None, n_lon=None, params: QdParams | None = None, use_ocean=None, quiet=False, device=0):
        env = os.environ
        n_lat = int(n_lat if n_lat is not None else env.get("QD_N_LAT", "121"))   # run_simulation.py:1195 is 121x240
        n_lon = int(n_lon if n_lon is not None else env.get("QD_N_LON", "240"))
        self.quiet = quiet
        self.grid = SphericalGrid(n_lat, n_lon)
        self.land_mask, self.elevation = topo.create_land_sea_mask(self.grid, return_elevation=True)
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
        use_ocean = (int(env.get("QD_USE_OCEAN", "1")) == 1) if use_ocean is None else bool(use_ocean)
        self.ocean = None
        if use_ocean:
            H_ocean = float(env.get("QD_OCEAN_H_M", str(H_mld)))
            self.ocean = WindDrivenSlabOcean(self.grid, self.land_mask, H_ocean,
                                             init_Ts=np.where(self.land_mask == 0, 288.0, 288.0))
        self.forcing = ThermalForcing(self.grid, OrbitalSystem())
        self.t = 0.0
        self.dt = int(env.get("QD_DT_SECONDS", "300"))
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
        save_restart(path, self.grid, self.dev, self.t, self.land_mask)

    # -- the loop
    def run_steps(self, n):
        """n iterations of run_simulation.py:1760-2340, one resident qd_step_n call."""
        if n <= 0:
            return
        times = self.t + self.dt * np.arange(n)
        stars = self.forcing.star_table(times)
        self.dev.step_n(stars, float(self.dt), with_ocean=self.ocean is not None, with_physics=True, pass_albedo=False,
                        with_hydrology=True)
        self.t = float(times[-1] + self.dt)

    def diagnostics(self):
        d = self.dev
        from ._lib import R_MAXABS, R_COSMEAN
        return {"max|u|": d.reduce("U", R_MAXABS), "max|v|": d.reduce("V", R_MAXABS), "max|h|": d.reduce("H", R_MAXABS),
                "<T_s>": d.reduce("TS", R_COSMEAN), "<cloud>": d.reduce("CLOUD", R_COSMEAN),
                "<E>": d.reduce("EFLUX", R_COSMEAN), "<P>": d.reduce("PRECIP", R_COSMEAN)}


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
    restart_in = env.get("QD_RESTART_IN")
    if restart_in and os.path.exists(restart_in):
        sim.load(restart_in)
        print(f"[Restart] loaded '{restart_in}' at t={sim.t:.1f} s")
    elif sim.t == 0.0:
        if env.get("QD_ORBIT_EPOCH_SECONDS"):
            sim.t = float(env["QD_ORBIT_EPOCH_SECONDS"])
        elif env.get("QD_ORBIT_EPOCH_DAYS"):
            sim.t = float(env["QD_ORBIT_EPOCH_DAYS"]) * day
    t0 = sim.t
    n_total = len(np.arange(t0, t0 + duration, sim.dt))
    print(f"Grid resolution: {sim.grid.n_lat} lat x {sim.grid.n_lon} lon | dt = {sim.dt} s | "
          f"{duration / day:.1f} planetary days | {n_total} steps")
    print("[Plots] matplotlib panels are not produced by the device driver (out of the hot path).")

    autosave_on = int(env.get("QD_AUTOSAVE_ENABLE", "1")) == 1
    restart_out = env.get("QD_RESTART_OUT") or os.path.join("data", "restart_autosave.nc")
    every_h = float(env.get("QD_ECO_AUTOSAVE_EVERY_HOURS", "0") or 0)
    autosave_steps = int(every_h * 3600 / sim.dt) if every_h > 0 else 0
    state = {"saved": False}

    def _autosave(reason):
        if state["saved"] or not autosave_on:
            return
        try:
            sim.save(restart_out)
            print(f"[Autosave] ({reason}) wrote {restart_out} at t={sim.t:.1f} s")
        except Exception as e:     # noqa: BLE001  (the reference never lets I/O kill the run)
            print(f"[Autosave] skipped: {e}")
        state["saved"] = True

    def _on_signal(signum, _frame):
        _autosave("signal")
        sys.exit(130 if signum == signal.SIGINT else 143)
    signal.signal(signal.SIGINT, _on_signal)
    signal.signal(signal.SIGTERM, _on_signal)
    atexit.register(lambda: _autosave("atexit"))

    chunk = max(1, min(200, autosave_steps or 200))
    done = 0
    wall0 = time.perf_counter()
    while done < n_total:
        n = min(chunk, n_total - done)
        sim.run_steps(n)
        done += n
        if int(env.get("QD_DYN_DIAG_PRINT", "1")) == 1:
            dg = sim.diagnostics()
            el = time.perf_counter() - wall0
            print(f"t={sim.t / day:8.2f} d | " + " ".join(f"{k}={v:.4g}" for k, v in dg.items()) +
                  f" | {done / max(el, 1e-9):.0f} steps/s")
        if autosave_steps and done % autosave_steps == 0:
            state["saved"] = False
            _autosave("periodic")
            state["saved"] = False
    if env.get("QD_RESTART_OUT"):
        sim.save(env["QD_RESTART_OUT"])
        print(f"[Restart] wrote {env['QD_RESTART_OUT']}")
        state["saved"] = True
    print("--- Simulation Finished ---")
    return 0


if __name__ == "__main__":
    sys.exit(main())
