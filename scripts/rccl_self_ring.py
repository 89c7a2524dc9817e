#!/usr/bin/env python3
"""One latitude band of an N-band decomposition on ONE GPU over the real RCCL transport: a communicator of one rank whose ring
neighbours are the rank itself (its halos are refreshed with its own edge rows, so the physics is not meaningful -- the timing is:
per-band kernel time + RCCL call overheads, without xGMI latency).  Lower bound for the per-rank step time of `bench.py --gpus N`.

    python scripts/rccl_self_ring.py [n_lat n_lon n_bands steps [halo]]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_PORT", "29733")
import ctypes  # noqa: E402

import numpy as np  # noqa: E402

import qingdai_amd as qa  # noqa: E402
from qingdai_amd.bands import band_ranges, init_rccl, required_halo  # noqa: E402
from qingdai_amd.device import Device  # noqa: E402
from qingdai_amd.topography import create_land_sea_mask, generate_base_properties  # noqa: E402

nlat, nlon, nb, nsteps = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (721, 1440, 8, 60)))
HALO = int(sys.argv[5]) if len(sys.argv) > 5 else None
TRANSPORTS = ("rccl",) if HALO else ("local", "rccl")
grid = qa.SphericalGrid(nlat, nlon)
mask = create_land_sea_mask(grid)
alb, fric = generate_base_properties(mask)
p = qa.QdParams(energy_w=1.0, cloud_couple=1)
p.has_csmap = 0
forcing = qa.ThermalForcing(qa.SphericalGrid(nlat, nlon), qa.OrbitalSystem())
stars = forcing.star_table([i * 300.0 for i in range(nsteps)])
r0, n = band_ranges(nlat, nb)[nb // 2]
for transport in TRANSPORTS:
    dev = Device(qa.SphericalGrid(nlat, nlon), p, row0=r0, n_rows=n, halo=HALO or required_halo(nlat), rank=0, world=1)
    if transport == "local":
        arr = (ctypes.c_void_p * 1)(dev.h)
        assert dev.lib.qd_comm_init_local(arr, 1) == 0
    else:
        init_rccl(dev, 0, 1, tag="selfring")
    for k, v in {"LAND_MASK": mask, "FRICTION": fric, "BASE_ALBEDO": alb}.items():
        dev.upload_now(k, v)
    dev.step_n(stars[:12], 300.0, with_ocean=True, with_physics=True, pass_albedo=True)
    dev.sync()
    ne0, na0 = ctypes.c_int(0), ctypes.c_int(0)
    dev.lib.qd_comm_stats(dev.h, ctypes.byref(ne0)); dev.lib.qd_comm_allreduce_count(dev.h, ctypes.byref(na0))
    t0 = time.perf_counter()
    dev.step_n(stars[12:], 300.0, with_ocean=True, with_physics=True, pass_albedo=True)
    dev.sync()
    el = time.perf_counter() - t0
    ne, na, nh = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    dev.lib.qd_comm_stats(dev.h, ctypes.byref(ne)); dev.lib.qd_comm_allreduce_count(dev.h, ctypes.byref(na))
    dev.lib.qd_comm_host_allreduce_count(dev.h, ctypes.byref(nh))
    k = nsteps - 12
    print(f"{transport:5s}: band rows {r0}..{r0 + n} of {nlat} (1/{nb}), halo {HALO or required_halo(nlat)}: {el / k * 1e3:.3f} ms/step, "
          f"{(ne.value - ne0.value) / k:.1f} halo exchanges + {(na.value - na0.value) / k:.1f} RCCL all-reduces per step "
          f"(+ {nh.value / nsteps:.1f} through the host ring), n_sub {dev.last_ocean_nsub()}")
    dev.close()
