"""
qingdai_amd/_lib.py -- ctypes binding of libqingdai_hip.so (include/qingdai_hip.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible the
product raises.  (The oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes
import os

from .params import qd_params

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QD_LIB_PATH") or os.path.join(_HERE, "libqingdai_hip.so")     # QD_LIB_PATH: developer A/B builds

# field ids (enum qd_field)
FIELDS = ["U", "V", "H", "TS", "Q", "CLOUD", "HICE", "ISR", "ISR_A", "ISR_B", "TEQ", "ALBEDO",
          "OLR", "EFLUX", "PCOND", "LH", "LHREL", "CLOUD_EFF", "FRICTION", "CSMAP", "BASE_ALBEDO", "ELEVATION",
          "UO", "VO", "ETA", "SST", "QNET", "PRECIP", "CLOUD_FROM_P", "CLOUD_SRC", "W_LAND", "S_SNOW", "C_SNOW",
          "S_SNOW_NEXT", "MELT", "P_RAIN", "GLACIER", "RUNOFF",
          "ECO_LAI", "ECO_LAI_SNAP", "ECO_F", "ECO_EDAY", "ECO_ALPHA", "ECO_ALPHA_BANDED", "WATER_ALPHA"]
F = {n: i for i, n in enumerate(FIELDS)}
F["LAND_MASK"] = 100
F["ICE_MASK"] = 101
R_SUM, R_COSMEAN, R_MAX, R_MIN, R_MAXABS = 0, 1, 2, 3, 4

# every symbol include/qingdai_hip.h declares
SYMBOLS = [
    "qd_abi_version", "qd_device_count", "qd_create", "qd_destroy", "qd_last_error", "qd_upload", "qd_download", "qd_set_params",
    "qd_get_step_counter", "qd_set_step_counter", "qd_forcing", "qd_simple_albedo", "qd_atmos_step",
    "qd_ocean_step", "qd_driver_physics", "qd_hydrology_commit", "qd_step_n", "qd_last_ocean_nsub", "qd_sync",
    "qd_op_laplacian", "qd_op_hyperdiffuse", "qd_op_advect", "qd_op_shapiro", "qd_op_zonal_filter", "qd_op_divergence",
    "qd_op_vorticity", "qd_op_gaussian", "qd_op_median_positive", "qd_median_state", "qd_reduce", "qd_energy_diagnostics", "qd_energy_diagnostics_last", "qd_band_insolation",
    "qd_comm_unique_id", "qd_comm_init", "qd_comm_init_local", "qd_comm_stats", "qd_comm_allreduce_count", "qd_comm_grouped_sum_count", "qd_comm_init_shm", "qd_comm_host_allreduce_count", "qd_hostring_open", "qd_hostring_allreduce",
    "qd_hostring_close", "qd_comm_barrier", "qd_comm_allreduce_max", "qd_peer_export", "qd_peer_connect", "qd_comm_peer_stats", "qd_comm_peer_carried", "qd_peer_selftest", "qd_peer_disable", "qd_tune_reload",
    "qd_plansim_create", "qd_plansim_destroy", "qd_plansim_plan", "qd_plansim_mark", "qd_plansim_margin", "qd_plansim_segments",
    "qd_plansim_pop_exchange", "qd_plansim_segments_rows",
    "qd_eco_configure", "qd_eco_set_lai_layers", "qd_eco_substep", "qd_eco_banded_alpha", "qd_eco_get_state", "qd_eco_set_state",
    "qd_indiv_configure", "qd_indiv_substep", "qd_indiv_download", "qd_indiv_upload",
    "qd_phyto_configure", "qd_phyto_upload", "qd_phyto_download", "qd_phyto_advect_diffuse",
    "qd_copy_ceiling", "qd_timing_enable", "qd_timing_select", "qd_timing_get", "qd_timing_reset",
]


class qd_eco_params(ctypes.Structure):
    """include/qingdai_hip.h: qd_eco_params"""
    _fields_ = ([(n, ctypes.c_double) for n in ("k_canopy", "leaf_scalar", "soil_ref", "w_lai", "light_update_hours",
                                                 "recompute_lai_delta")] +
                [(n, ctypes.c_int32) for n in ("substep_every_nphys", "albedo_couple", "bands_couple", "water_couple", "use_lai", "map_f32")])


class qd_grid_desc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("n_lat", "n_lon", "row0", "n_rows", "halo", "device", "rank", "world")]


_lib = None


class QdError(RuntimeError):
    pass


def load():
    """Load the HIP library; fail loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QdError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"or `make -C qingdai_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    vp, dp, i32, i64, dbl, sz = (ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int64,
                                 ctypes.c_double, ctypes.c_size_t)
    lib.qd_abi_version.restype = i32
    lib.qd_device_count.restype = i32
    lib.qd_create.argtypes = [ctypes.POINTER(qd_grid_desc), ctypes.POINTER(qd_params), dbl, ctypes.POINTER(vp)]
    lib.qd_destroy.argtypes = [vp]
    lib.qd_last_error.argtypes = [vp]
    lib.qd_last_error.restype = ctypes.c_char_p
    lib.qd_upload.argtypes = [vp, i32, vp, sz]
    lib.qd_download.argtypes = [vp, i32, vp, sz]
    lib.qd_set_params.argtypes = [vp, ctypes.POINTER(qd_params), sz]
    lib.qd_get_step_counter.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    lib.qd_set_step_counter.argtypes = [vp, i64, i64]
    lib.qd_forcing.argtypes = [vp, dp, dp, dbl, i32]
    lib.qd_simple_albedo.argtypes = [vp, dbl]
    lib.qd_atmos_step.argtypes = [vp, dbl, i32]
    lib.qd_ocean_step.argtypes = [vp, dbl, i32, i32, i32]
    lib.qd_driver_physics.argtypes = [vp, dbl]
    lib.qd_hydrology_commit.argtypes = [vp, dbl]
    lib.qd_step_n.argtypes = [vp, i32, dbl, i32, dp]
    lib.qd_last_ocean_nsub.argtypes = [vp, ctypes.POINTER(i32)]
    lib.qd_sync.argtypes = [vp]
    lib.qd_op_laplacian.argtypes = [vp, vp, i32, vp]
    lib.qd_op_hyperdiffuse.argtypes = [vp, vp, vp, dbl, dbl, i32, i32, vp]
    lib.qd_op_advect.argtypes = [vp, vp, vp, vp, dbl, i32, vp]
    lib.qd_op_shapiro.argtypes = [vp, vp, i32, vp]
    lib.qd_op_zonal_filter.argtypes = [vp, vp, dbl, dbl, vp]
    lib.qd_op_divergence.argtypes = [vp, vp, vp, vp]
    lib.qd_op_vorticity.argtypes = [vp, vp, vp, vp]
    lib.qd_op_gaussian.argtypes = [vp, vp, dbl, i32, vp]
    lib.qd_op_median_positive.argtypes = [vp, vp, dbl, dp]
    lib.qd_reduce.argtypes = [vp, i32, i32, dp]
    lib.qd_energy_diagnostics.argtypes = [vp, dp]
    lib.qd_energy_diagnostics_last.argtypes = [vp, dp]
    lib.qd_copy_ceiling.argtypes = [vp, sz, i32, dp]
    lib.qd_band_insolation.argtypes = [vp, i32, dp, dp, dp, vp]
    ip = ctypes.POINTER(ctypes.c_int32)
    lib.qd_eco_configure.argtypes = [vp, ctypes.POINTER(qd_eco_params), sz]
    lib.qd_eco_set_lai_layers.argtypes = [vp, vp, i32, i32]
    lib.qd_eco_substep.argtypes = [vp, dbl]
    lib.qd_eco_banded_alpha.argtypes = [vp, i32, dp, dp]
    lib.qd_eco_get_state.argtypes = [vp, dp]
    lib.qd_eco_set_state.argtypes = [vp, dp]
    lib.qd_indiv_configure.argtypes = [vp, i32, ip, ip, i32, ip, vp, vp, i32, dp, dp, dp, i32, dbl, dbl, i32]
    lib.qd_indiv_substep.argtypes = [vp, dbl, ip]
    lib.qd_indiv_download.argtypes = [vp, vp, vp]
    lib.qd_indiv_upload.argtypes = [vp, vp, vp]
    lib.qd_phyto_configure.argtypes = [vp, i32, dbl, dbl]
    lib.qd_phyto_upload.argtypes = [vp, i32, vp]
    lib.qd_phyto_download.argtypes = [vp, i32, vp]
    lib.qd_phyto_advect_diffuse.argtypes = [vp, dbl]
    lib.qd_comm_unique_id.argtypes = [vp, sz]
    lib.qd_comm_init.argtypes = [vp, vp, sz]
    lib.qd_comm_barrier.argtypes = [vp]
    lib.qd_comm_init_local.argtypes = [ctypes.POINTER(vp), i32]
    lib.qd_median_state.argtypes = [vp, dp]
    lib.qd_comm_stats.argtypes = [vp, ctypes.POINTER(i32)]
    lib.qd_comm_allreduce_count.argtypes = [vp, ctypes.POINTER(i32)]
    lib.qd_comm_grouped_sum_count.argtypes = [vp, ctypes.POINTER(i32)]
    lib.qd_comm_init_shm.argtypes = [vp, ctypes.c_char_p]
    lib.qd_comm_host_allreduce_count.argtypes = [vp, ctypes.POINTER(i32)]
    lib.qd_hostring_open.argtypes = [ctypes.c_char_p, i32, i32, ctypes.POINTER(vp)]
    lib.qd_hostring_allreduce.argtypes = [vp, dp, i32, i32]
    lib.qd_hostring_close.argtypes = [vp]
    lib.qd_comm_allreduce_max.argtypes = [vp, dp, i32]
    lib.qd_tune_reload.argtypes = [vp]
    lib.qd_peer_export.argtypes = [vp, vp, sz]
    lib.qd_peer_connect.argtypes = [vp, vp, sz, i32]
    lib.qd_peer_selftest.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_longlong)]
    lib.qd_peer_disable.argtypes = [vp]
    lib.qd_comm_peer_stats.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    lib.qd_comm_peer_carried.argtypes = [vp]
    ip = ctypes.POINTER(i32)
    lib.qd_plansim_create.argtypes = [ctypes.POINTER(qd_grid_desc), ctypes.POINTER(vp)]
    lib.qd_plansim_destroy.argtypes = [vp]
    lib.qd_plansim_plan.argtypes = [vp, ip, ip, i32, i32]
    lib.qd_plansim_mark.argtypes = [vp, ip, i32, i32]
    lib.qd_plansim_margin.argtypes = [vp, i32]
    lib.qd_plansim_segments.argtypes = [vp, i32, ip]
    lib.qd_plansim_pop_exchange.argtypes = [vp, ip, i32, ip]
    lib.qd_plansim_segments_rows.argtypes = [vp, i32, i32, ip]
    lib.qd_timing_enable.argtypes = [vp, i32]
    lib.qd_timing_select.argtypes = [vp, ctypes.c_char_p]
    lib.qd_timing_get.argtypes = [vp, ctypes.c_char_p, dp, ctypes.POINTER(i64)]
    lib.qd_timing_reset.argtypes = [vp]
    for s in SYMBOLS:
        fn = getattr(lib, s)
        if s not in ("qd_last_error",):
            fn.restype = i32
    lib.qd_last_error.restype = ctypes.c_char_p
    _lib = lib
    return lib
