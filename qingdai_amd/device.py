"""
qingdai_amd/device.py -- one resident device context (qd_handle) per grid.

The reference keeps atmosphere and ocean state in two Python objects that exchange
NumPy arrays every step (run_simulation.py:2194-2253).  Here both live in one device
context so the coupling never leaves HBM; `SpectralModel` and `WindDrivenSlabOcean`
are views onto it.  Attribute reads download, attribute writes upload lazily
(DoubleBufferingArray contract, numerics/double_buffer.py:47-184).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import F, QdError
from .params import QdParams


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


class Device:
    def __init__(self, grid, params: QdParams | None = None, device=0, row0=0, n_rows=None, halo=0, rank=0, world=1):
        self.lib = _lib.load()
        self.grid = grid
        self.params = params or QdParams.from_env()
        self.shape = (grid.n_lat, grid.n_lon)
        n_rows = grid.n_lat if n_rows is None else n_rows
        desc = _lib.qd_grid_desc(grid.n_lat, grid.n_lon, row0, n_rows, halo, device, rank, world)
        h = ctypes.c_void_p()
        ps = self.params.to_struct()
        rc = self.lib.qd_create(ctypes.byref(desc), ctypes.byref(ps), float(self.params.q_init_rh), ctypes.byref(h))
        if rc != 0:
            raise QdError("qd_create failed: " + (self.lib.qd_last_error(None) or b"?").decode())
        self.h = h
        self._host = {}           # field -> ndarray handed to the caller (may have been mutated)
        self._dirty = set()       # fields whose host copy is newer than the device copy
        grid._device = self

    # ---- errors
    def _chk(self, rc, what):
        if rc != 0:
            raise QdError(f"{what} failed: " + (self.lib.qd_last_error(self.h) or b"?").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.qd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters
    def push_params(self):
        ps = self.params.to_struct()
        self._chk(self.lib.qd_set_params(self.h, ctypes.byref(ps), ctypes.sizeof(ps)), "qd_set_params")

    # ---- attribute surface
    def get(self, name):
        """Download (or return the cached host copy of) a field."""
        if name in self._host:
            return self._host[name]
        if name in ("LAND_MASK", "ICE_MASK"):
            out = np.empty(self.shape, dtype=np.uint8)
        else:
            out = np.empty(self.shape, dtype=np.float64)
        self._chk(self.lib.qd_download(self.h, F[name], out.ctypes.data, out.nbytes), f"qd_download({name})")
        self._host[name] = out
        return out

    def set(self, name, arr):
        if name in ("LAND_MASK", "ICE_MASK"):
            a = _c(arr, np.uint8)
        else:
            a = _c(np.broadcast_to(np.asarray(arr, dtype=np.float64), self.shape))
            if not a.flags.writeable:
                a = a.copy()
        if a.shape != self.shape:
            raise ValueError(f"{name}: expected shape {self.shape}, got {a.shape}")
        self._host[name] = a
        self._dirty.add(name)

    def flush(self):
        """Send every host-side copy that may differ from the device (anything written, or
        read and possibly mutated in place) before device work starts."""
        for name in list(self._host):
            a = self._host[name]
            self._chk(self.lib.qd_upload(self.h, F[name], a.ctypes.data, a.nbytes), f"qd_upload({name})")
        self._host.clear()
        self._dirty.clear()

    def upload_now(self, name, arr):
        self.set(name, arr)
        a = self._host.pop(name)
        self._dirty.discard(name)
        self._chk(self.lib.qd_upload(self.h, F[name], a.ctypes.data, a.nbytes), f"qd_upload({name})")

    # ---- the path
    def forcing(self, star_a, star_b, theta, with_teq=True):
        self.flush()
        A = (ctypes.c_double * 3)(*star_a)
        B = (ctypes.c_double * 3)(*star_b)
        self._chk(self.lib.qd_forcing(self.h, A, B, float(theta), 1 if with_teq else 0), "qd_forcing")

    def simple_albedo(self, ocean_albedo=0.08):
        self.flush()
        self._chk(self.lib.qd_simple_albedo(self.h, float(ocean_albedo)), "qd_simple_albedo")

    def atmos_step(self, dt, has_albedo):
        self.flush()
        self._chk(self.lib.qd_atmos_step(self.h, float(dt), 1 if has_albedo else 0), "qd_atmos_step")

    def ocean_step(self, dt, compute_qnet, use_ice_mask, inject_sst):
        self.flush()
        self._chk(self.lib.qd_ocean_step(self.h, float(dt), int(compute_qnet), int(use_ice_mask), int(inject_sst)),
                  "qd_ocean_step")

    def driver_physics(self, dt):
        self.flush()
        self._chk(self.lib.qd_driver_physics(self.h, float(dt)), "qd_driver_physics")

    def hydrology_commit(self, dt):
        self.flush()
        self._chk(self.lib.qd_hydrology_commit(self.h, float(dt)), "qd_hydrology_commit")

    def step_n(self, stars, dt, with_ocean=False, with_physics=False, pass_albedo=True, with_hydrology=False, energy_diag=False,
               ecology=False, phyto=False):
        """benchmark_jax.py:124-158 as one resident loop (qd_step_n).  `stars`: [n][7] host
        scalars from ThermalForcing.star_table()."""
        self.flush()
        st = np.ascontiguousarray(stars, dtype=np.float64)
        assert st.ndim == 2 and st.shape[1] == 7
        flags = ((1 if with_ocean else 0) | (2 if with_physics else 0) | (4 if pass_albedo else 0) | (8 if with_hydrology else 0) |
                 (16 if energy_diag else 0) | (32 if ecology else 0) | (64 if phyto else 0))
        self._chk(self.lib.qd_step_n(self.h, int(st.shape[0]), float(dt), flags,
                                     st.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), "qd_step_n")

    def sync(self):
        self._chk(self.lib.qd_sync(self.h), "qd_sync")

    # ---- phytoplankton tracers carried by the ocean currents (pygcm/ecology/phyto.py:496-547), resident
    def phyto_configure(self, n_species, K_h, adv_alpha):
        self._chk(self.lib.qd_phyto_configure(self.h, int(n_species), float(K_h), float(adv_alpha)), "qd_phyto_configure")
        self._phyto_S = int(n_species)

    def phyto_upload(self, C_s):
        C_s = np.ascontiguousarray(C_s, dtype=np.float64)
        assert C_s.ndim == 3 and C_s.shape[0] == getattr(self, "_phyto_S", -1) and C_s.shape[1:] == self.shape
        for s in range(C_s.shape[0]):
            self._chk(self.lib.qd_phyto_upload(self.h, s, C_s[s].ctypes.data), "qd_phyto_upload")

    def phyto_download(self):
        self.flush()
        out = np.zeros((self._phyto_S,) + self.shape, dtype=np.float64)
        for s in range(self._phyto_S):
            self._chk(self.lib.qd_phyto_download(self.h, s, out[s].ctypes.data), "qd_phyto_download")
        return out

    def phyto_advect_diffuse(self, dt_seconds):
        """PhytoManager.advect_diffuse on the resident tracers and the resident uo / vo (three launches for all species)."""
        self.flush()
        self._chk(self.lib.qd_phyto_advect_diffuse(self.h, float(dt_seconds)), "qd_phyto_advect_diffuse")

    def last_ocean_nsub(self):
        n = ctypes.c_int(0)
        self.lib.qd_last_ocean_nsub(self.h, ctypes.byref(n))
        return n.value

    def counters(self):
        a, o = ctypes.c_int64(0), ctypes.c_int64(0)
        self.lib.qd_get_step_counter(self.h, ctypes.byref(a), ctypes.byref(o))
        return a.value, o.value

    def set_counters(self, a, o):
        self.lib.qd_set_step_counter(self.h, int(a), int(o))

    def reduce(self, name, op):
        self.flush()
        out = ctypes.c_double(0.0)
        self._chk(self.lib.qd_reduce(self.h, F[name], int(op), ctypes.byref(out)), "qd_reduce")
        return out.value

    # ---- timing hooks
    def timing(self, on=True, select=None):
        if select:
            self.lib.qd_timing_select(self.h, select.encode())
        else:
            self.lib.qd_timing_enable(self.h, 1 if on else 0)
        self.lib.qd_timing_reset(self.h)

    def timing_get(self, name):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        self.lib.qd_timing_get(self.h, name.encode(), ctypes.byref(ms), ctypes.byref(n))
        return ms.value, n.value

    # ---- operator seam (jax_compat.py:111-216): host in, host out
    def _out(self):
        return np.empty(self.shape, dtype=np.float64)

    def op_laplacian(self, Fh, ocean=False):
        a, out = _c(Fh), self._out()
        self._chk(self.lib.qd_op_laplacian(self.h, a.ctypes.data, 1 if ocean else 0, out.ctypes.data), "qd_op_laplacian")
        return out

    def op_hyperdiffuse(self, Fh, k4, dt, n_substeps=1, ocean=False):
        a, out = _c(Fh), self._out()
        if np.isscalar(k4):
            rc = self.lib.qd_op_hyperdiffuse(self.h, a.ctypes.data, None, float(k4), float(dt), int(n_substeps),
                                             1 if ocean else 0, out.ctypes.data)
        else:
            k = np.asarray(k4, dtype=np.float64)
            if k.ndim == 2:          # the reference's maps are constant along longitude
                if not np.all(k == k[:, :1]):
                    raise ValueError("k4 map must be a function of latitude only")
                k = k[:, 0]
            k = _c(k)
            rc = self.lib.qd_op_hyperdiffuse(self.h, a.ctypes.data, k.ctypes.data, 0.0, float(dt), int(n_substeps),
                                             1 if ocean else 0, out.ctypes.data)
        self._chk(rc, "qd_op_hyperdiffuse")
        return out

    def op_advect(self, field, u, v, dt, ocean=False):
        a, uu, vv, out = _c(field), _c(u), _c(v), self._out()
        self._chk(self.lib.qd_op_advect(self.h, a.ctypes.data, uu.ctypes.data, vv.ctypes.data, float(dt),
                                        1 if ocean else 0, out.ctypes.data), "qd_op_advect")
        return out

    def op_shapiro(self, Fh, n=2):
        a, out = _c(Fh), self._out()
        self._chk(self.lib.qd_op_shapiro(self.h, a.ctypes.data, int(n), out.ctypes.data), "qd_op_shapiro")
        return out

    ENERGY_DIAG_KEYS = ("TOA_net", "SFC_net", "ATM_net", "I_mean", "R_mean", "OLR_mean", "SW_sfc_mean", "LW_sfc_mean",
                        "SH_mean", "LH_mean")

    def energy_diagnostics(self):
        """energy.compute_energy_diagnostics (energy.py:494-538) of the resident state -> dict of global means."""
        self.flush()
        out = (ctypes.c_double * 10)()
        self._chk(self.lib.qd_energy_diagnostics(self.h, out), "qd_energy_diagnostics")
        return dict(zip(self.ENERGY_DIAG_KEYS, [float(x) for x in out]))

    def copy_ceiling(self, nbytes=1 << 30, reps=8):
        """Measured device-to-device streaming rate in GB/s (read + written bytes), past the Infinity Cache by default."""
        out = ctypes.c_double(0.0)
        self._chk(self.lib.qd_copy_ceiling(self.h, int(nbytes), int(reps), ctypes.byref(out)), "qd_copy_ceiling")
        return out.value

    def energy_diagnostics_last(self):
        """The means taken inside the last step_n(..., energy_diag=True): first step, after time_step, before the ocean."""
        out = (ctypes.c_double * 10)()
        self._chk(self.lib.qd_energy_diagnostics_last(self.h, out), "qd_energy_diagnostics_last")
        return dict(zip(self.ENERGY_DIAG_KEYS, [float(x) for x in out]))

    def op_zonal_filter(self, Fh, cutoff=0.75, damp=0.5):
        """SpectralModel._spectral_zonal_filter (dynamics.py:233-258)."""
        a, out = _c(Fh), self._out()
        self._chk(self.lib.qd_op_zonal_filter(self.h, a.ctypes.data, float(cutoff), float(damp), out.ctypes.data), "qd_op_zonal_filter")
        return out

    def op_divvort(self, u, v, vort=False):
        uu, vv, out = _c(u), _c(v), self._out()
        fn = self.lib.qd_op_vorticity if vort else self.lib.qd_op_divergence
        self._chk(fn(self.h, uu.ctypes.data, vv.ctypes.data, out.ctypes.data), "qd_op_div/vort")
        return out

    def op_gaussian(self, Fh, sigma, mode="reflect"):
        a, out = _c(Fh), self._out()
        self._chk(self.lib.qd_op_gaussian(self.h, a.ctypes.data, float(sigma), 1 if mode == "wrap" else 0,
                                          out.ctypes.data), "qd_op_gaussian")
        return out

    def op_median_positive(self, x, default):
        a = _c(x)
        out = ctypes.c_double(0.0)
        self._chk(self.lib.qd_op_median_positive(self.h, a.ctypes.data, float(default), ctypes.byref(out)),
                  "qd_op_median_positive")
        return out.value
