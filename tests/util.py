"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np

import qd_oracle as qo

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
STATE = ("u", "v", "h", "T_s", "q", "cloud_cover", "h_ice")
DIAG = ("E_flux_last", "P_cond_flux_last", "LH_last", "LH_release_last", "olr")


def load_golden(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.loads(str(d["meta"]))
    return meta, d


def relerr(a, b):
    """max |a-b| / max|b| : error relative to the field's scale (stencil results cancel)."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b))) / max(float(np.max(np.abs(b))), 1e-300)


def surface(nlat, nlon):
    from qingdai_amd.topography import create_land_sea_mask, generate_base_properties
    g = qo.Grid(nlat, nlon)
    mask = create_land_sea_mask(g)
    alb, fric = generate_base_properties(mask)
    return g, mask, alb, fric


def oracle_params(over):
    return qo.defaults(**over)


def run_oracle_time_step(meta, d):
    nlat, nlon = meta["nlat"], meta["nlon"]
    g, mask, alb, fric = surface(nlat, nlon)
    P = oracle_params(meta["over"])
    m = qo.AtmosOracle(g, fric, mask, P, C_s_map=np.where(mask == 1, 3e6, P.Cs_ocean).astype(float))
    for k in STATE:
        setattr(m, k, d["init_" + k].copy())
    f = qo.Forcing(g)
    albedo = np.where(mask == 0, 0.08, alb)
    dt = meta["dt"]
    for i in range(meta["nsteps"]):
        t = i * dt
        a_, b_ = f.insolation_components(t)
        m.isr_A, m.isr_B, m.isr = a_, b_, a_ + b_
        Teq = f.equilibrium_temp(t, albedo)
        m.time_step(Teq, dt, albedo=albedo if meta["with_albedo"] else None)
    return m


def product_params(over):
    from qingdai_amd import QdParams
    return QdParams(**over)


def run_device_time_step(meta, d, resident_forcing=True):
    """The same benchmark_jax-style loop through the HIP path (C-ABI via ctypes)."""
    import qingdai_amd as qa
    nlat, nlon = meta["nlat"], meta["nlon"]
    _, mask, alb, fric = surface(nlat, nlon)
    grid = qa.SphericalGrid(nlat, nlon)
    p = product_params(meta["over"])
    Cs_ocean = 1000.0 * 4200.0 * 50.0
    m = qa.SpectralModel(grid, fric, H=8000, tau_rad=10 * 24 * 3600, greenhouse_factor=0.40,
                         C_s_map=np.where(mask == 1, 3e6, Cs_ocean).astype(float), land_mask=mask,
                         Cs_ocean=Cs_ocean, Cs_land=3e6, Cs_ice=5e6, params=p)
    for k in STATE:
        setattr(m, k, d["init_" + k].copy())
    forcing = qa.ThermalForcing(grid, qa.OrbitalSystem())
    albedo = np.where(mask == 0, 0.08, alb)
    dt = meta["dt"]
    m._dev.set("ALBEDO", albedo)
    for i in range(meta["nsteps"]):
        t = i * dt
        if resident_forcing:
            forcing.update_device(t, with_teq=True)
            m.time_step(None, dt, albedo=True if meta["with_albedo"] else None)
        else:
            insA, insB = forcing.calculate_insolation_components(t)
            m.isr_A, m.isr_B, m.isr = insA, insB, insA + insB
            Teq = forcing.calculate_equilibrium_temp(t, albedo)
            m.time_step(Teq, dt, albedo=albedo if meta["with_albedo"] else None)
    return m
