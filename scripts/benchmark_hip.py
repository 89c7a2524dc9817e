#!/usr/bin/env python3
"""The reference's timing harness scripts/benchmark_jax.py (run_benchmark, :43-162) against the HIP path:
same CLI (--nlat --nlon --steps --dt --with-ocean), same loop (2 warm-up steps, timed steps, prints
per-step wall time and simulated days)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import qingdai_amd as qa  # noqa: E402
from qingdai_amd.topography import create_land_sea_mask, generate_base_properties  # noqa: E402


def run_benchmark(nlat, nlon, steps, dt, with_ocean):
    print(f"[Benchmark] Backend: HIP/gfx950 | Grid: {nlat}x{nlon}, steps={steps}, dt={dt}s, with_ocean={with_ocean}")
    grid = qa.SphericalGrid(nlat, nlon)
    mask = create_land_sea_mask(grid)
    alb, fric = generate_base_properties(mask)
    Cs_ocean = float(os.getenv("QD_RHO_W", "1000")) * float(os.getenv("QD_CP_W", "4200")) * float(os.getenv("QD_MLD_M", "50"))
    gcm = qa.SpectralModel(grid, fric, H=8000, tau_rad=10 * 24 * 3600, greenhouse_factor=float(os.getenv("QD_GH_FACTOR", "0.40")),
                           C_s_map=np.where(mask == 1, 3e6, Cs_ocean).astype(float), land_mask=mask, Cs_ocean=Cs_ocean,
                           Cs_land=float(os.getenv("QD_CS_LAND", "3e6")), Cs_ice=float(os.getenv("QD_CS_ICE", "5e6")))
    if with_ocean:
        qa.WindDrivenSlabOcean(grid, mask, float(os.getenv("QD_OCEAN_H_M", "50")), init_Ts=np.full((nlat, nlon), 288.0))
    gcm._dev.upload_now("BASE_ALBEDO", alb)
    forcing = qa.ThermalForcing(grid, qa.OrbitalSystem())
    dev = gcm._dev
    warm = min(2, max(0, steps // 10))
    if warm:
        dev.step_n(forcing.star_table([0.0] * warm), dt, with_ocean=with_ocean, pass_albedo=True)
    dev.sync()
    t0 = time.perf_counter()
    dev.step_n(forcing.star_table([i * dt for i in range(steps)]), dt, with_ocean=with_ocean, pass_albedo=True)
    dev.sync()
    t1 = time.perf_counter()
    per = (t1 - t0) / max(1, steps)
    print(f"[Benchmark] Total wall time: {t1 - t0:.3f} s | per-step: {per:.6f} s | "
          f"sim_days={steps * dt / (2 * np.pi / 8.726646259971648e-5):.3f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nlat", type=int, default=121)
    ap.add_argument("--nlon", type=int, default=240)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--dt", type=float, default=300.0)
    ap.add_argument("--with-ocean", action="store_true", default=False)
    a = ap.parse_args()
    print("=== Qingdai GCM HIP Benchmark ===")
    run_benchmark(a.nlat, a.nlon, a.steps, a.dt, a.with_ocean)
