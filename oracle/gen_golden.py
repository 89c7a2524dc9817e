#!/usr/bin/env python3
"""
oracle/gen_golden.py -- TEST INFRASTRUCTURE ONLY.  Runs ONLY in the authoring
container, where the reference is mounted read-only at /root/reference.

Imports the reference (pygcm.*) from /root/reference, runs its per-timestep path
on small seeded cases and writes inputs + the REFERENCE's outputs as .npz
fixtures under tests/golden/.  At the same time it runs the oracle restatement
(oracle/qd_oracle) on the same inputs and prints the max deviation, so a
regeneration doubles as the oracle-vs-reference validation.

Nothing of the reference's source is written anywhere: fixtures hold arrays and
the parameter values of each case only.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden.py
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("QD_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.dont_write_bytecode = True
OUT = os.path.join(REPO, "tests", "golden")

import qd_oracle as qo                                   # noqa: E402
from qd_oracle import atmos as oat, numerics as onx, physics as oph, column as ocol  # noqa: E402
from qingdai_amd.topography import create_land_sea_mask, generate_base_properties  # noqa: E402

# env var name for every oracle parameter the cases below override
ENV_OF = {
    "energy_w": "QD_ENERGY_W", "mom_scheme": "QD_MOM_SCHEME", "filter_type": "QD_FILTER_TYPE",
    "cloud_couple": "QD_CLOUD_COUPLE", "lw_v2": "QD_LW_V2", "gh_lock": "QD_GH_LOCK",
    "seaice_enabled": "QD_USE_SEAICE", "shapiro_every": "QD_SHAPIRO_EVERY", "shapiro_n": "QD_SHAPIRO_N",
    "spec_every": "QD_SPEC_EVERY", "diff_q": "QD_DIFF_Q", "diff_cloud": "QD_DIFF_CLOUD",
    "k4_nsub": "QD_K4_NSUB", "pcond_ref": "QD_PCOND_REF", "tau_cond": "QD_TAU_COND",
    "ocean_outlier": "QD_OCEAN_OUTLIER", "ocean_shapiro_n": "QD_OCEAN_SHAPIRO_N",
    "ocean_shapiro_every": "QD_OCEAN_SHAPIRO_EVERY", "K_h": "QD_KH_OCEAN", "ocean_cfl": "QD_OCEAN_CFL",
    "k4_u": "QD_K4_U", "k4_q": "QD_K4_Q", "diff_every": "QD_DIFF_EVERY", "gh_factor_lw": "QD_GH_FACTOR",
    "diff_factor": "QD_DIFF_FACTOR", "ocean_ice_qfac": "QD_OCEAN_ICE_QFAC",
}


@contextlib.contextmanager
def ref_env(over):
    """Export the case's overrides as the QD_* variables the reference reads."""
    saved = {k: v for k, v in os.environ.items() if k.startswith("QD_")}
    for k in list(os.environ):
        if k.startswith("QD_"):
            del os.environ[k]
    os.environ["QD_ENERGY_DIAG"] = "0"
    os.environ["QD_OCEAN_ENERGY_DIAG"] = "0"
    os.environ["QD_USE_JAX"] = "0"
    for k, v in over.items():
        name = ENV_OF[k]
        if k == "mom_scheme":
            v = "primitive" if v == 1 else "geos"
        os.environ[name] = str(v)
    try:
        yield
    finally:
        for k in list(os.environ):
            if k.startswith("QD_"):
                del os.environ[k]
        os.environ.update(saved)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def maxrel(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    s = max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b))) / s


def save(name, meta, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print(f"  wrote {os.path.relpath(path, REPO)} ({os.path.getsize(path) / 1024:.1f} KiB)")


def surface(nlat, nlon):
    from pygcm.grid import SphericalGrid
    g = SphericalGrid(nlat, nlon)
    mask = create_land_sea_mask(g)
    alb, fric = generate_base_properties(mask)
    return g, mask, alb, fric


STATE = ("u", "v", "h", "T_s", "q", "cloud_cover", "h_ice")
DIAG = ("E_flux_last", "P_cond_flux_last", "LH_last", "LH_release_last", "olr")


def perturbed_state(shape, seed, wet=False, cloudy=False, icy=False):
    """Seeded non-trivial initial state so every branch sees structure."""
    r = np.random.default_rng(seed)
    nlat, nlon = shape
    lat = np.linspace(-np.pi / 2, np.pi / 2, nlat)[:, None]
    lon = np.linspace(0, 2 * np.pi, nlon)[None, :]
    st = {}
    st["u"] = 25.0 * np.cos(lat) * np.sin(2 * lon) + r.normal(0, 3.0, shape)
    st["v"] = 8.0 * np.sin(2 * lat) * np.cos(3 * lon) + r.normal(0, 2.0, shape)
    st["h"] = 8000.0 + 300 * np.sin(lat) ** 2 + 40.0 * np.cos(lat) * np.cos(2 * lon) + r.normal(0, 2.0, shape)
    st["T_s"] = 262.0 + 38.0 * np.cos(lat) ** 2 + r.normal(0, 1.0, shape)
    st["q"] = np.clip(0.006 + 0.004 * np.cos(lat) ** 2 + r.normal(0, 5e-4, shape), 0.0, 0.5)
    st["cloud_cover"] = np.zeros(shape)
    st["h_ice"] = np.zeros(shape)
    if wet:       # super-saturate so condensation + the P_cond median branch fire
        st["h"] = st["h"] - 7600.0
        st["q"] = st["q"] + 0.02 * (r.random(shape) > 0.5)
    if cloudy:
        st["cloud_cover"] = np.clip(0.3 + 0.3 * np.sin(3 * lon) * np.cos(lat) + r.normal(0, 0.05, shape), 0, 1)
    if icy:
        st["h_ice"] = np.where(np.abs(lat) > 1.1, 0.4 + 0.3 * r.random(shape), 0.0) * np.ones(shape)
        st["T_s"] = np.where(np.abs(lat) > 1.1, 268.0 + 4 * r.random(shape), st["T_s"])
    return st


def build_ref_model(g, mask, fric, st=None):
    from pygcm.dynamics import SpectralModel
    Cs_ocean = 1000.0 * 4200.0 * 50.0
    C_s_map = np.where(mask == 1, 3e6, Cs_ocean).astype(float)
    m = SpectralModel(g, fric, H=8000, tau_rad=10 * 24 * 3600, greenhouse_factor=0.40, C_s_map=C_s_map,
                      land_mask=mask, Cs_ocean=Cs_ocean, Cs_land=3e6, Cs_ice=5e6)
    if st is not None:
        for k in STATE:
            setattr(m, k, st[k].copy())
    return m


def build_oracle_model(g, mask, fric, P, st=None):
    og = qo.Grid(g.n_lat, g.n_lon)
    C_s_map = np.where(mask == 1, 3e6, P.Cs_ocean).astype(float)
    m = qo.AtmosOracle(og, fric, mask, P, C_s_map=C_s_map)
    if st is not None:
        for k in STATE:
            setattr(m, k, st[k].copy())
    return og, m


# --------------------------------------------------------------------- cases
def case_operators(nlat, nlon, seed):
    """Every operator of SURVEY section 2 (O1-O6, O11, O12) in isolation."""
    from pygcm.ocean import WindDrivenSlabOcean
    from scipy.ndimage import gaussian_filter
    g, mask, alb, fric = surface(nlat, nlon)
    st = perturbed_state((nlat, nlon), seed)
    F = st["h"]
    dt = 300.0
    with ref_env({}):
        m = build_ref_model(g, mask, fric, st)
        oc = WindDrivenSlabOcean(g, mask, 50.0)
        cos3 = np.maximum(np.cos(np.deg2rad(g.lat_mesh)), 1e-3)
        k4 = 0.02 * np.minimum(m.a * m.dlat_rad, m.a * m.dlon_rad * cos3) ** 4 / dt
        ref = dict(
            lap_atm=m._laplacian_sphere(F.copy()),
            lap_ocn=oc._laplacian_sphere(F.copy()),
            hyper_atm=m._hyperdiffuse(F.copy(), k4, dt, n_substeps=1),
            hyper_atm_nsub2=m._hyperdiffuse(F.copy(), 0.5 * k4, dt, n_substeps=2),
            hyper_scalar=m._hyperdiffuse(F.copy(), 1.0e14, dt, n_substeps=1),
            shapiro2=m._shapiro_filter(F.copy(), n=2),
            shapiro1=m._shapiro_filter(F.copy(), n=1),
            spectral=m._spectral_zonal_filter(F.copy(), 0.75, 0.5),
            advect_atm=m._advect(st["T_s"].copy(), dt),
            advect_ocn=oc._advect_scalar(st["T_s"].copy(), 0.02 * st["u"], 0.02 * st["v"], dt),
            div=g.divergence(st["u"], st["v"]),
            vort=g.vorticity(st["u"], st["v"]),
            grad_lon=np.gradient(F, m.dlon_rad, axis=1),
            grad_lat=np.gradient(F, m.dlat_rad, axis=0),
            gauss1=gaussian_filter(F, sigma=1.0),
            gauss02_wrap=gaussian_filter(st["T_s"], sigma=0.2, mode="wrap"),
        )
    # storm-force winds: polar rows travel 1e4-1e5 cells before the fold
    r = np.random.default_rng(seed + 7)
    ub = r.normal(0, 60.0, F.shape)
    vb = r.normal(0, 40.0, F.shape)
    with ref_env({}):
        m.u, m.v = ub.copy(), vb.copy()
        ref["advect_storm"] = m._advect(st["T_s"].copy(), dt)
    P = qo.defaults()
    og = qo.Grid(nlat, nlon)
    cosl = np.cos(np.deg2rad(og.lat_mesh))
    c02, c05, c6 = np.maximum(cosl, 0.2), np.maximum(cosl, 0.5), np.maximum(1e-6, cosl)
    dl, dn, a = og.dlat_rad, og.dlon_rad, P.a
    orc = dict(
        lap_atm=oat.laplacian_sphere(F, dl, dn, c02, a),
        lap_ocn=oat.laplacian_sphere(F, dl, dn, c05, a),
        hyper_atm=oat.hyperdiffuse(F, k4, dt, 1, dl, dn, c02, a),
        hyper_atm_nsub2=oat.hyperdiffuse(F, 0.5 * k4, dt, 2, dl, dn, c02, a),
        hyper_scalar=oat.hyperdiffuse(F, 1.0e14, dt, 1, dl, dn, c02, a),
        shapiro2=onx.shapiro(F, 2), shapiro1=onx.shapiro(F, 1),
        spectral=oat.spectral_zonal_filter(F, 0.75, 0.5, nlon),
        advect_atm=oat.advect_semilag(st["T_s"], st["u"], st["v"], dt, a, dl, dn, c6),
        advect_ocn=oat.advect_semilag(st["T_s"], 0.02 * st["u"], 0.02 * st["v"], dt, a, dl, dn, c05),
        advect_storm=oat.advect_semilag(st["T_s"], ub, vb, dt, a, dl, dn, c6),
        div=og.divergence(st["u"], st["v"]), vort=og.vorticity(st["u"], st["v"]),
        grad_lon=onx.gradient_axis1(F, dn), grad_lat=onx.gradient_axis0(F, dl),
        gauss1=onx.gaussian_filter(F, 1.0), gauss02_wrap=onx.gaussian_filter(st["T_s"], 0.2, "wrap"),
    )
    for k in ref:
        print(f"    {k:18s} oracle-vs-ref maxrel {maxrel(orc[k], ref[k]):.2e}  bitexact={np.array_equal(orc[k], ref[k])}")
    save(f"ops_{nlat}x{nlon}", dict(kind="operators", nlat=nlat, nlon=nlon, seed=seed, dt=dt),
         F=F, T=st["T_s"], u=st["u"], v=st["v"], k4=k4, u_storm=ub, v_storm=vb,
         **{"ref_" + k: v for k, v in ref.items()})


def case_time_step(name, nlat, nlon, nsteps, over, with_albedo, seed=None, wet=False, cloudy=False, icy=False,
                   save_every=None):
    """Whole SpectralModel.time_step through the benchmark_jax.py:124-132 loop."""
    from pygcm.forcing import ThermalForcing
    from pygcm.orbital import OrbitalSystem
    g, mask, alb, fric = surface(nlat, nlon)
    st = perturbed_state((nlat, nlon), seed, wet, cloudy, icy) if seed is not None else None
    dt = 300.0
    albedo = np.where(mask == 0, 0.08, alb)
    snaps = {}
    with ref_env(over):
        forcing = ThermalForcing(g, OrbitalSystem())
        m = build_ref_model(g, mask, fric, st)
        init = {k: getattr(m, k).copy() for k in STATE}
        for i in range(nsteps):
            t = i * dt
            insA, insB = forcing.calculate_insolation_components(t)
            m.isr_A, m.isr_B = insA, insB
            m.isr = insA + insB
            Teq = forcing.calculate_equilibrium_temp(t, albedo)
            if with_albedo:
                m.time_step(Teq, dt, albedo=albedo)
            else:
                m.time_step(Teq, dt)
            if save_every and ((i + 1) % save_every == 0 or i == 0):
                for k in STATE:
                    snaps[f"s{i + 1}_{k}"] = getattr(m, k).copy()
    P = qo.defaults(**over)
    og, om = build_oracle_model(g, mask, fric, P, st)
    of = qo.Forcing(og)
    for i in range(nsteps):
        t = i * dt
        a_, b_ = of.insolation_components(t)
        om.isr_A, om.isr_B, om.isr = a_, b_, a_ + b_
        Teq_o = of.equilibrium_temp(t, albedo)
        om.time_step(Teq_o, dt, albedo=albedo if with_albedo else None)
    worst = 0.0
    for k in STATE + DIAG:
        d = maxrel(getattr(om, k), getattr(m, k))
        worst = max(worst, d)
        print(f"    {k:18s} oracle-vs-ref maxrel {d:.2e}")
    if with_albedo:
        print(f"    cloud_eff_last     oracle-vs-ref maxrel {maxrel(om.cloud_eff_last, m.cloud_eff_last):.2e}")
    print(f"    Teq (forcing)      oracle-vs-ref maxrel {maxrel(Teq_o, Teq):.2e}")
    arrays = {"init_" + k: v for k, v in init.items()}
    arrays.update({"ref_" + k: getattr(m, k) for k in STATE + DIAG})
    if with_albedo:
        arrays["ref_cloud_eff_last"] = m.cloud_eff_last
    arrays["ref_Teq_last"] = Teq
    arrays["ref_isr_last"] = m.isr
    arrays.update(snaps)
    save(name, dict(kind="time_step", nlat=nlat, nlon=nlon, nsteps=nsteps, dt=dt, over=over,
                    with_albedo=with_albedo, seed=seed, save_every=save_every), **arrays)
    return worst


def case_ocean(name, nlat, nlon, nsteps, over, seed, strong=False, poison=False):
    """WindDrivenSlabOcean.step driven by fixed winds / Q_net / ice mask."""
    from pygcm.ocean import WindDrivenSlabOcean
    g, mask, alb, fric = surface(nlat, nlon)
    st = perturbed_state((nlat, nlon), seed, icy=True)
    r = np.random.default_rng(seed + 3)
    dt = 300.0
    u_atm = st["u"] * (4.0 if strong else 1.0)
    v_atm = st["v"] * (4.0 if strong else 1.0)
    Q_net = 120.0 * np.cos(np.deg2rad(g.lat_mesh)) - 60.0 + r.normal(0, 5.0, mask.shape)
    ice = st["h_ice"] > 0.0
    init_Ts = np.where(mask == 0, st["T_s"], 288.0)
    with ref_env(over):
        oc = WindDrivenSlabOcean(g, mask, 50.0, init_Ts=init_Ts)
        # seed non-zero currents / eta so every term is live from step 1
        oc.uo = np.where(mask == 0, 0.3 * np.cos(np.deg2rad(g.lat_mesh)) + r.normal(0, 0.05, mask.shape), 0.0)
        oc.vo = np.where(mask == 0, r.normal(0, 0.05, mask.shape), 0.0)
        oc.eta = np.where(mask == 0, r.normal(0, 0.2, mask.shape), 0.0)
        if poison:      # NaN in every prognostic field (ocean cells, one next to a pole): exercises the step's nan_to_num placements
            oi, oj = np.where(mask == 0)
            pick = [int(len(oi) * f) for f in (0.13, 0.41, 0.67, 0.93)]
            oc.uo[oi[pick[0]], oj[pick[0]]] = np.nan
            oc.vo[oi[pick[1]], oj[pick[1]]] = np.nan
            oc.eta[oi[pick[2]], oj[pick[2]]] = np.nan
            oc.Ts[oi[pick[3]], oj[pick[3]]] = np.nan
            init_Ts = oc.Ts.copy()
        init = dict(uo=oc.uo.copy(), vo=oc.vo.copy(), eta=oc.eta.copy(), Ts=oc.Ts.copy())
        nsubs = []
        for i in range(nsteps):
            # recompute n_sub the way step() does, for the record
            oc.step(dt, u_atm, v_atm, Q_net=Q_net, ice_mask=ice)
    P = qo.defaults(**over)
    og = qo.Grid(nlat, nlon)
    oo = qo.OceanOracle(og, mask, P, init_Ts=init_Ts)
    oo.uo, oo.vo, oo.eta = init["uo"].copy(), init["vo"].copy(), init["eta"].copy()
    for i in range(nsteps):
        oo.step(dt, u_atm, v_atm, Q_net=Q_net, ice_mask=ice)
        nsubs.append(oo.last_n_sub)
    for k in ("uo", "vo", "eta", "Ts"):
        print(f"    {k:18s} oracle-vs-ref maxrel {maxrel(getattr(oo, k), getattr(oc, k)):.2e}")
    print(f"    n_sub per step: {nsubs}")
    save(name, dict(kind="ocean", nlat=nlat, nlon=nlon, nsteps=nsteps, dt=dt, over=over, seed=seed, n_sub=nsubs),
         u_atm=u_atm, v_atm=v_atm, Q_net=Q_net, ice_mask=ice.astype(np.uint8),
         **{"init_" + k: v for k, v in init.items()},
         **{"ref_" + k: getattr(oc, k) for k in ("uo", "vo", "eta", "Ts")})


def case_physics(nlat, nlon, seed):
    """Driver-side diagnostics (physics.py) + forcing on a seeded state."""
    from types import SimpleNamespace
    from pygcm import physics as rph
    from pygcm.forcing import ThermalForcing
    from pygcm.orbital import OrbitalSystem
    g, mask, alb, fric = surface(nlat, nlon)
    st = perturbed_state((nlat, nlon), seed, cloudy=True, icy=True)
    r = np.random.default_rng(seed + 11)
    Pc_dry = np.zeros(mask.shape)
    Pc_wet = np.where(r.random(mask.shape) > 0.6, 1e-5 * r.random(mask.shape), 0.0)
    ice_frac = 1.0 - np.exp(-np.maximum(st["h_ice"], 0.0) / 0.5)
    out = {}
    with ref_env({}):
        for tag, Pc in (("dry", Pc_dry), ("wet", Pc_wet)):
            ns = SimpleNamespace(u=st["u"], v=st["v"], T_s=st["T_s"], cloud_cover=st["cloud_cover"], P_cond_flux_last=Pc)
            out[f"precip_{tag}"] = rph.diagnose_precipitation_hybrid(ns, g, D_crit=-1e-7, k_precip=1e5, beta_div=0.4)
        ns = SimpleNamespace(u=st["u"], v=st["v"], T_s=st["T_s"], cloud_cover=st["cloud_cover"], P_cond_flux_last=Pc_dry)
        out["precip_legacy"] = rph.diagnose_precipitation(ns, g, -1e-7, 1e5)
        pref = float(np.median(out["precip_dry"][out["precip_dry"] > 0]))
        out["cloud_from_p"] = rph.cloud_from_precip(out["precip_dry"], C_max=0.95, P_ref=pref)
        out["cloud_source"] = rph.parameterize_cloud_cover(ns, g, mask)
        out["albedo"] = rph.calculate_dynamic_albedo(st["cloud_cover"], st["T_s"], alb, 0.6, 0.5, land_mask=mask, ice_frac=ice_frac)
        out["albedo_T"] = rph.calculate_dynamic_albedo(st["cloud_cover"], st["T_s"], alb, 0.6, 0.5, land_mask=mask)
        forcing = ThermalForcing(g, OrbitalSystem())
        for j, t in enumerate((0.0, 12345.0, 3.3e7)):
            a_, b_ = forcing.calculate_insolation_components(t)
            out[f"isrA_{j}"], out[f"isrB_{j}"] = a_, b_
            out[f"Teq_{j}"] = forcing.calculate_equilibrium_temp(t, out["albedo"])
    P = qo.defaults()
    og = qo.Grid(nlat, nlon)
    orc = {}
    for tag, Pc in (("dry", Pc_dry), ("wet", Pc_wet)):
        ns = SimpleNamespace(u=st["u"], v=st["v"], T_s=st["T_s"], cloud_cover=st["cloud_cover"], P_cond_flux_last=Pc)
        orc[f"precip_{tag}"] = oph.diagnose_precipitation_hybrid(ns, og, P)
    ns = SimpleNamespace(u=st["u"], v=st["v"], T_s=st["T_s"], cloud_cover=st["cloud_cover"], P_cond_flux_last=Pc_dry)
    orc["precip_legacy"] = oph.diagnose_precipitation(ns, og, -1e-7, 1e5)
    orc["cloud_from_p"] = oph.cloud_from_precip(orc["precip_dry"], 0.95, float(np.median(orc["precip_dry"][orc["precip_dry"] > 0])))
    orc["cloud_source"] = oph.parameterize_cloud_cover(ns, og)
    orc["albedo"] = oph.calculate_dynamic_albedo(st["cloud_cover"], st["T_s"], alb, 0.6, 0.5, land_mask=mask, ice_frac=ice_frac)
    orc["albedo_T"] = oph.calculate_dynamic_albedo(st["cloud_cover"], st["T_s"], alb, 0.6, 0.5, land_mask=mask)
    of = qo.Forcing(og)
    for j, t in enumerate((0.0, 12345.0, 3.3e7)):
        a_, b_ = of.insolation_components(t)
        orc[f"isrA_{j}"], orc[f"isrB_{j}"] = a_, b_
        orc[f"Teq_{j}"] = of.equilibrium_temp(t, orc["albedo"])
    for k in out:
        print(f"    {k:18s} oracle-vs-ref maxrel {maxrel(orc[k], out[k]):.2e}")
    save(f"physics_{nlat}x{nlon}", dict(kind="physics", nlat=nlat, nlon=nlon, seed=seed, times=[0.0, 12345.0, 3.3e7]),
         u=st["u"], v=st["v"], T_s=st["T_s"], cloud_cover=st["cloud_cover"], h_ice=st["h_ice"],
         Pc_wet=Pc_wet, base_albedo=alb, **{"ref_" + k: v for k, v in out.items()})


def case_orography(nlat, nlon, seed):
    """compute_orographic_factor (physics.py:116-161) and the hybrid precipitation with that factor
    (run_simulation.py:1769-1781), from the REFERENCE's functions on a seeded elevation map."""
    from types import SimpleNamespace
    from pygcm import physics as rph
    from qingdai_amd.topography import generate_elevation_map
    g, mask, alb, fric = surface(nlat, nlon)
    st = perturbed_state((nlat, nlon), seed, cloudy=True, icy=False)
    r = np.random.default_rng(seed + 100)
    elev = np.maximum(generate_elevation_map(g, seed=seed), 0.0) * (mask == 1)
    Pc = np.abs(r.normal(3e-5, 2e-5, (nlat, nlon)))
    with ref_env({}):
        fac = rph.compute_orographic_factor(g, elev, st["u"], st["v"], k_orog=7e-4)
        ns = SimpleNamespace(u=st["u"], v=st["v"], T_s=st["T_s"], cloud_cover=st["cloud_cover"], P_cond_flux_last=Pc)
        pr = rph.diagnose_precipitation_hybrid(ns, g, D_crit=-1e-7, k_precip=1e5, orog_factor=fac, smooth_sigma=1.0,
                                               beta_div=0.4, renorm=True)
    P = qo.defaults(orog_enable=1)
    og = qo.Grid(nlat, nlon)
    ofac = oph.compute_orographic_factor(og, elev, st["u"], st["v"], k_orog=7e-4)
    opr = oph.diagnose_precipitation_hybrid(ns, og, P, orog_factor=ofac)
    print(f"    orog_factor        oracle-vs-ref maxrel {maxrel(ofac, fac):.2e}   (max factor {fac.max():.3f})")
    print(f"    precip_orog        oracle-vs-ref maxrel {maxrel(opr, pr):.2e}")
    # a strong QD_OROG_K so the cap (2.0) and the [1, 3] clip inside the precipitation are exercised
    K2 = 0.08
    with ref_env({}):
        fac2 = rph.compute_orographic_factor(g, elev, st["u"], st["v"], k_orog=K2)
        pr2 = rph.diagnose_precipitation_hybrid(ns, g, D_crit=-1e-7, k_precip=1e5, orog_factor=fac2, smooth_sigma=1.0,
                                                beta_div=0.4, renorm=True)
    ofac2 = oph.compute_orographic_factor(og, elev, st["u"], st["v"], k_orog=K2)
    opr2 = oph.diagnose_precipitation_hybrid(ns, og, P, orog_factor=ofac2)
    print(f"    strong k: factor maxrel {maxrel(ofac2, fac2):.2e} (max {fac2.max():.3f}); precip maxrel {maxrel(opr2, pr2):.2e}")
    save(f"orog_{nlat}x{nlon}", dict(kind="orography", nlat=nlat, nlon=nlon, seed=seed, k_orog=7e-4, k_orog_strong=K2),
         u=st["u"], v=st["v"], elevation=elev, Pc=Pc, T_s=st["T_s"], cloud_cover=st["cloud_cover"],
         ref_orog_factor=fac, ref_precip_orog=pr, ref_orog_factor_strong=fac2, ref_precip_orog_strong=pr2)


def case_phyto(nlat, nlon, seed):
    """PhytoManager.advect_diffuse (pygcm/ecology/phyto.py:496-547) run as the REFERENCE's own method, bound to a
    minimal stand-in object carrying exactly the attributes the method reads."""
    import types
    from types import SimpleNamespace
    from pygcm.ecology.phyto import PhytoManager
    from qd_oracle import phyto as ophy
    g, mask, alb, fric = surface(nlat, nlon)
    r = np.random.default_rng(seed)
    S = 3
    lat = np.deg2rad(g.lat_mesh)
    C = np.abs(r.normal(0.3, 0.2, (S, nlat, nlon))) * (mask == 0) * (0.5 + np.cos(lat) ** 2)
    uo = 0.4 * np.cos(lat) * np.sin(2 * np.deg2rad(g.lon_mesh)) + r.normal(0, 0.05, (nlat, nlon))
    vo = 0.2 * np.sin(2 * lat) * np.cos(3 * np.deg2rad(g.lon_mesh)) + r.normal(0, 0.05, (nlat, nlon))
    dt = 300.0
    fake = SimpleNamespace(S=S, land_mask=mask, C_phyto_s=C.copy(), K_h=5.0e3, NL=nlat, NM=nlon, a=6.371e6 if False else None,
                           dlat=g.dlat_rad, dlon=g.dlon_rad, coslat=np.maximum(np.cos(lat), 0.5))
    from pygcm import constants as rconst
    fake.a = rconst.PLANET_RADIUS
    fake._advect_scalar = types.MethodType(PhytoManager._advect_scalar, fake)
    fake._laplacian_sphere = types.MethodType(PhytoManager._laplacian_sphere, fake)
    with ref_env({}):
        for _ in range(3):
            PhytoManager.advect_diffuse(fake, uo, vo, dt)
    out = C.copy()
    og = qo.Grid(nlat, nlon)
    for _ in range(3):
        out = ophy.advect_diffuse(out, uo, vo, dt, og, mask, K_h=5.0e3, adv_alpha=0.7)
    print(f"    phyto C after 3 steps   oracle-vs-ref maxrel {maxrel(out, fake.C_phyto_s):.2e}")
    save(f"phyto_{nlat}x{nlon}", dict(kind="phyto", nlat=nlat, nlon=nlon, seed=seed, dt=dt, nsteps=3, K_h=5.0e3, adv_alpha=0.7),
         C0=C, uo=uo, vo=vo, ref_C=fake.C_phyto_s)


def case_spectral_bands(nlat, nlon):
    """dual_star_insolation_to_bands (pygcm/ecology/spectral.py:304-426) from the REFERENCE, NB = 16 (default mode) and NB = 8
    in Rayleigh mode, on the reference's own two-star insolation at two times."""
    from pygcm.ecology import spectral as rsp
    from pygcm.forcing import ThermalForcing
    from pygcm.orbital import OrbitalSystem
    from qd_oracle import spectral as osp
    g, mask, alb, fric = surface(nlat, nlon)
    out = {}
    with ref_env({}):
        forcing = ThermalForcing(g, OrbitalSystem())
        for ti, t in enumerate((0.0, 4.1e6)):
            insA, insB = forcing.calculate_insolation_components(t)
            out[f"insA_{ti}"], out[f"insB_{ti}"] = insA, insB
            out[f"ref_bands16_{ti}"] = rsp.dual_star_insolation_to_bands(insA, insB, rsp.make_bands(16, 380.0, 780.0))
        os.environ["QD_ECO_TOA_TO_SURF_MODE"] = "rayleigh"
        out["ref_bands8_rayleigh_1"] = rsp.dual_star_insolation_to_bands(out["insA_1"], out["insB_1"], rsp.make_bands(8, 400.0, 700.0))
        b16 = rsp.make_bands(16, 380.0, 780.0)
        out["ref_specA16"] = rsp.blackbody_band_weights(rsp.estimate_teff_from_LM(0.7, 0.914, j=0.8), b16)
        out["ref_specB16"] = rsp.blackbody_band_weights(rsp.estimate_teff_from_LM(0.410, 0.8, j=0.8), b16)
    for ti in (0, 1):
        o = osp.dual_star_insolation_to_bands(out[f"insA_{ti}"], out[f"insB_{ti}"], osp.make_bands(16, 380.0, 780.0))
        print(f"    bands16 t{ti}          oracle-vs-ref maxrel {maxrel(o, out[f'ref_bands16_{ti}']):.2e}")
    o = osp.dual_star_insolation_to_bands(out["insA_1"], out["insB_1"], osp.make_bands(8, 400.0, 700.0), rayleigh=dict(mode="rayleigh"))
    print(f"    bands8 rayleigh      oracle-vs-ref maxrel {maxrel(o, out['ref_bands8_rayleigh_1']):.2e}")
    save(f"spectral_{nlat}x{nlon}", dict(kind="spectral", nlat=nlat, nlon=nlon, times=[0.0, 4.1e6]), **out)


def case_ecology(nlat, nlon, seed, variant="default"):
    """The per-step ecology of BASELINE config 5 run as the REFERENCE's own classes: EcologyAdapter.step_subdaily over
    PopulationManager (pygcm/ecology/adapter.py:140-186, population.py:252-294,831-915), get_surface_albedo_bands
    (population.py:875-893) with the driver's daily reduction (run_simulation.py:1843-1844), the driver's base-albedo blend
    (run_simulation.py:2075-2141, composed here around the reference's calculate_dynamic_albedo), and
    IndividualPool.try_substep (individuals.py:142-191).  The LAI layers are perturbed between steps the way step_daily would."""
    from types import SimpleNamespace
    from pygcm import physics as rph
    from pygcm.ecology.adapter import EcologyAdapter
    from pygcm.ecology.individuals import IndividualPool
    from pygcm.forcing import ThermalForcing
    from pygcm.orbital import OrbitalSystem
    from pygcm import constants as rconst
    from qd_oracle import ecology as oeco, spectral as osp
    g, mask, alb, fric = surface(nlat, nlon)
    r = np.random.default_rng(seed)
    env = {"QD_ECO_DIAG": "0", "QD_ECO_NS": "4", "QD_ECO_COHORT_K": "2", "QD_ECO_LIGHT_UPDATE_EVERY_HOURS": "0.5",
           "QD_ECO_SUBSTEP_EVERY_NPHYS": "2", "QD_ECO_LAI_K": "0.6", "QD_ECO_SOIL_REFLECT": "0.18",
           "QD_ECO_INDIV_SAMPLE_FRAC": "0.3", "QD_ECO_INDIV_PER_CELL": "5", "QD_ECO_INDIV_SUBSTEPS_PER_DAY": "10"}
    NB, LAM, MODE, K_CAN, SOIL, EVERY_N = 16, (380.0, 780.0), "simple", 0.6, 0.18, 2
    if variant == "rayleigh":
        # Rayleigh band weighting, 8 bands over 400-700 nm, per-species genes from the environment, default canopy constants,
        # alpha on every step
        env = {"QD_ECO_DIAG": "0", "QD_ECO_NS": "3", "QD_ECO_COHORT_K": "1", "QD_ECO_LIGHT_UPDATE_EVERY_HOURS": "0.5",
               "QD_ECO_TOA_TO_SURF_MODE": "rayleigh", "QD_ECO_SPECTRAL_BANDS": "8", "QD_ECO_SPECTRAL_RANGE_NM": "400,700",
               "QD_ECO_SPECIES_0_PEAKS": "500:30:0.9", "QD_ECO_SPECIES_1_PEAKS": "440:25:0.5, 660:20:0.7, 550:0:0.4",
               "QD_ECO_SPECIES_1_DROUGHT_TOL": "0.6", "QD_ECO_SPECIES_WEIGHTS": "0.5,0.3,0.2",
               "QD_ECO_INDIV_SAMPLE_FRAC": "0.3", "QD_ECO_INDIV_PER_CELL": "5", "QD_ECO_INDIV_SUBSTEPS_PER_DAY": "10"}
        NB, LAM, MODE, K_CAN, SOIL, EVERY_N = 8, (400.0, 700.0), "rayleigh", 0.5, 0.20, 1
    RAY = dict(mode="rayleigh") if MODE == "rayleigh" else None
    dt, nsteps = 300.0, 10
    out, meta_steps = {}, []
    with ref_env({}):
        os.environ.update(env)
        forcing = ThermalForcing(g, OrbitalSystem())
        eco = quiet(EcologyAdapter, g, mask)
        pop = eco.pop
        S, K = pop.LAI_layers_SK.shape[:2]
        land = (mask == 1)
        L0 = np.abs(r.normal(0.4, 0.3, (S, K, nlat, nlon))) * land
        L0[0, 0][r.random((nlat, nlon)) < 0.03] = -0.05            # a few negative layers: max(LAI_tot, 0) matters
        L1 = L0 * (1.0 + 0.4 * r.random(L0.shape))                 # growth spurt: ratio >= delta, cache rebuilt
        L2 = L1 * (1.0 + 0.01 * r.normal(0, 1, L0.shape))          # small drift: ratio < delta, cache kept until the clock
        pop.LAI_layers_SK[...] = L0
        pop._lai_snapshot = pop.total_LAI().copy()
        n_rec = [0]
        orig = pop._recompute_canopy_cache

        def counted():
            n_rec[0] += 1
            orig()
        pop._recompute_canopy_cache = counted
        last_alpha = None
        for i in range(nsteps):
            if i == 2:
                pop.LAI_layers_SK[...] = L1
            if i == 5:
                pop.LAI_layers_SK[...] = L2
            insA, insB = forcing.calculate_insolation_components(i * dt * 40.0)
            out[f"insA_{i}"], out[f"insB_{i}"] = insA, insB
            a = eco.step_subdaily(insA + insB, 0.3, dt)
            meta_steps.append(dict(returned=a is not None, n_recompute=n_rec[0]))
            if a is not None:
                out[f"ref_alpha_{i}"] = a.copy()
                last_alpha = a.copy()
            if i == nsteps - 1:
                # the driver's blend (W_LAI = 0.8) + snow blend + dynamic albedo around the reference's own function
                glacier = land & (r.random((nlat, nlon)) < 0.2)
                C_snow = np.clip(r.random((nlat, nlon)) * 1.2 - 0.4, 0.0, 1.0) * land
                cloud = np.clip(r.random((nlat, nlon)), 0.0, 1.0)
                h_ice = np.maximum(r.normal(0.0, 0.3, (nlat, nlon)), 0.0) * (~land)
                Ts = 288.0 + r.normal(0, 10, (nlat, nlon))
                base_in = alb.copy()
                mm = land & (~glacier) & np.isfinite(last_alpha)
                base_in[mm] = (1.0 - 0.8) * base_in[mm] + 0.8 * last_alpha[mm]
                base_in[land] = np.clip((1.0 - C_snow[land]) * base_in[land] + C_snow[land] * 0.70, 0.0, 1.0)
                ice_frac = 1.0 - np.exp(-np.maximum(h_ice, 0.0) / 0.5)
                out["ref_albedo_blend"] = rph.calculate_dynamic_albedo(cloud, Ts, base_in, 0.6, 0.5, land_mask=mask, ice_frac=ice_frac)
                out.update(glacier=glacier.astype(np.uint8), C_snow=C_snow, cloud=cloud, h_ice=h_ice, Ts=Ts, land_mask=mask.astype(np.uint8), base_albedo=alb)
        out["ref_E_day"] = pop.E_day.copy()
        out["ref_f_cached"] = pop._canopy_f_cached.copy()
        A, w_b = eco.get_surface_albedo_bands()
        out["ref_A_bands"], out["ref_w_b"] = A, w_b
        out["ref_alpha_banded"] = np.clip(np.nansum(A * w_b[:, None, None], axis=0), 0.0, 1.0)
        out["R_species"] = pop._species_R_leaf.copy()
        out["species_w"] = pop.species_weights.copy()
        leaf_s = float(eco.alpha_leaf_scalar)
        # individuals: sampled pool, K = 10 substeps a day, driven with a long physics step so that several fire
        pool = quiet(IndividualPool, g, mask, eco)
        day = 2 * np.pi / rconst.PLANET_OMEGA
        dti = day / 25.0
        soil = np.clip(r.random((nlat, nlon)), 0.0, 1.0)
        fired = []
        for i in range(30):
            insA, insB = forcing.calculate_insolation_components(i * dti)
            before = pool._substep_accum if pool._substep_period is not None else 0.0
            pool.try_substep(insA, insB, eco, soil, dti, day)
            if pool._substep_accum < before + dti - 1e-9:
                fired.append(i)
                out[f"ind_insA_{i}"], out[f"ind_insB_{i}"] = insA, insB
        out.update(ind_sample_j=pool.sample_j, ind_sample_i=pool.sample_i, ind_cell=pool.indiv_cell_index, ind_Ab=pool.indiv_Ab,
                   ind_tol=pool.indiv_tol, ind_soil=soil, ref_ind_E_day=pool.indiv_E_day.copy(),
                   ref_ind_stress=pool.indiv_water_stress_days.copy())
    # the adapter without a population (QD_ECO_USE_LAI=0, adapter.py:79-80,162-166)
    with ref_env({}):
        os.environ.update(env)
        os.environ["QD_ECO_USE_LAI"] = "0"
        eco1 = quiet(EcologyAdapter, g, mask)
        m1 = [eco1.step_subdaily(out["insA_0"] + out["insB_0"], 0.3, dt) for _ in range(4)]
    pat = [(k + 1) % EVERY_N == 0 for k in range(4)]
    assert eco1.pop is None and [a is not None for a in m1] == pat
    out["ref_alpha_m1"] = m1[3]
    o1 = oeco.EcoAdapter(None, leaf_s, soil_ref=SOIL, substep_every_nphys=EVERY_N)
    got1 = [o1.step_subdaily(None, dt, land_mask=mask) for _ in range(4)]
    assert [a is not None for a in got1] == pat and np.array_equal(got1[3], m1[3], equal_nan=True)
    # oracle on the same inputs
    ob = osp.make_bands(NB, *LAM)
    opop = oeco.CanopyPopulation(mask, L0, k_canopy=K_CAN, light_update_every_hours=0.5, recompute_lai_delta=0.05)
    oad = oeco.EcoAdapter(opop, oeco.leaf_scalar(ob, MODE), soil_ref=SOIL, substep_every_nphys=EVERY_N)
    print(f"    leaf scalar           oracle-vs-ref {abs(oeco.leaf_scalar(ob, MODE) - leaf_s):.2e}")
    worst = 0.0
    for i in range(nsteps):
        if i == 2:
            opop.layers = L1.copy()
        if i == 5:
            opop.layers = L2.copy()
        a = oad.step_subdaily(out[f"insA_{i}"] + out[f"insB_{i}"], dt)
        assert (a is not None) == meta_steps[i]["returned"] and opop.n_recompute == meta_steps[i]["n_recompute"], (i, opop.n_recompute)
        if a is not None:
            assert np.array_equal(np.isnan(a), np.isnan(out[f"ref_alpha_{i}"]))
            worst = max(worst, maxrel(np.nan_to_num(a), np.nan_to_num(out[f"ref_alpha_{i}"])))
    print(f"    alpha maps            oracle-vs-ref maxrel {worst:.2e}   recomputes {opop.n_recompute}")
    print(f"    E_day / f cache       oracle-vs-ref maxrel {maxrel(opop.E_day, out['ref_E_day']):.2e} / {maxrel(opop.f_cached, out['ref_f_cached']):.2e}")
    R_eff = oeco.effective_leaf_reflectance(out["species_w"], out["R_species"])
    Ao = opop.surface_albedo_bands(R_eff, SOIL)
    print(f"    A_bands / banded      oracle-vs-ref maxrel {maxrel(np.nan_to_num(Ao), np.nan_to_num(A)):.2e} / "
          f"{maxrel(oeco.banded_alpha(Ao, oeco.band_weights(ob, MODE)), out['ref_alpha_banded']):.2e}")
    oi = oeco.IndividualSubstep(pool.sample_j, pool.sample_i, pool.indiv_cell_index, pool.indiv_Ab, pool.indiv_tol, 10)
    of = qo.Forcing(qo.Grid(nlat, nlon))
    ofired = []
    for i in range(30):
        a_, b_ = of.insolation_components(i * dti)
        if oi.try_substep(a_, b_, ob, soil, dti, day, rayleigh=RAY):
            ofired.append(i)
    assert ofired == fired, (ofired, fired)
    print(f"    individuals E / stress oracle-vs-ref maxrel {maxrel(oi.E_day, pool.indiv_E_day):.2e} / "
          f"{maxrel(oi.stress_days, pool.indiv_water_stress_days):.2e}   fired {len(fired)}")
    out["ind_species_tol"] = np.asarray([float(getattr(gg, "drought_tolerance", 0.5)) for gg in eco.genes_list])
    save(f"eco_{nlat}x{nlon}" + ("" if variant == "default" else "_" + variant),
         dict(kind="ecology", nlat=nlat, nlon=nlon, seed=seed, dt=dt, nsteps=nsteps, steps=meta_steps, k_canopy=K_CAN, soil_ref=SOIL,
              every_h=0.5, delta=0.05, substep_every=EVERY_N, leaf_scalar=leaf_s, w_lai=0.8, alpha_snow=0.70, ind_dt=dti, ind_day=day,
              ind_fired=fired, ind_k=10, nb=NB, lam=list(LAM), mode=MODE, env={k: v for k, v in env.items() if k != "QD_ECO_DIAG"}),
         L0=L0, L1=L1, L2=L2, **out)


def case_nonfinite(nlat, nlon, seed):
    """Where the reference scrubs non-finite values and where it lets them through: its own _laplacian_sphere (atmosphere and
    ocean), _hyperdiffuse, _shapiro_filter and a whole time_step on a state poisoned with NaN / +-inf."""
    from pygcm.ocean import WindDrivenSlabOcean
    g, mask, alb, fric = surface(nlat, nlon)
    st = perturbed_state((nlat, nlon), seed, cloudy=True)
    F = st["h"].copy()
    F[nlat // 2, 7] = np.nan; F[nlat - 2, nlon - 1] = np.inf; F[0, 3] = -np.inf
    dt = 300.0
    with ref_env({}), np.errstate(all="ignore"):
        m = build_ref_model(g, mask, fric, st)
        oc = WindDrivenSlabOcean(g, mask, 50.0)
        cos3 = np.maximum(np.cos(np.deg2rad(g.lat_mesh)), 1e-3)
        k4 = 0.02 * np.minimum(m.a * m.dlat_rad, m.a * m.dlon_rad * cos3) ** 4 / dt
        ref = dict(lap_atm=m._laplacian_sphere(F.copy()), lap_ocn=oc._laplacian_sphere(F.copy()),
                   hyper_atm=m._hyperdiffuse(F.copy(), k4, dt, n_substeps=1), shapiro2=m._shapiro_filter(F.copy(), n=2))
        st2 = {k: v.copy() for k, v in st.items()}
        st2["u"][nlat // 3, 11] = np.nan
        st2["q"][4, nlon - 2] = np.inf
        st2["cloud_cover"][nlat - 3, 2] = np.nan
        m2 = build_ref_model(g, mask, fric, st2)
        from pygcm.forcing import ThermalForcing
        from pygcm.orbital import OrbitalSystem
        forcing = ThermalForcing(g, OrbitalSystem())
        albedo = np.where(mask == 0, 0.08, alb)
        for i in range(2):
            insA, insB = forcing.calculate_insolation_components(i * dt)
            m2.isr_A, m2.isr_B, m2.isr = insA, insB, insA + insB
            m2.time_step(forcing.calculate_equilibrium_temp(i * dt, albedo), dt)
    P = qo.defaults()
    og = qo.Grid(nlat, nlon)
    cosl = np.cos(np.deg2rad(og.lat_mesh))
    with np.errstate(all="ignore"):
        orc = dict(lap_atm=oat.laplacian_sphere(F, og.dlat_rad, og.dlon_rad, np.maximum(cosl, 0.2), P.a),
                   lap_ocn=oat.laplacian_sphere(F, og.dlat_rad, og.dlon_rad, np.maximum(cosl, 0.5), P.a),
                   hyper_atm=oat.hyperdiffuse(F, k4, dt, 1, og.dlat_rad, og.dlon_rad, np.maximum(cosl, 0.2), P.a),
                   shapiro2=onx.shapiro(F, 2))
    for k in ref:
        same = np.array_equal(orc[k], ref[k], equal_nan=True)
        print(f"    {k:12s} oracle == reference (NaN-aware): {same}   nan {int(np.isnan(ref[k]).sum())} inf {int(np.isinf(ref[k]).sum())}")
    save(f"nonfinite_{nlat}x{nlon}", dict(kind="nonfinite", nlat=nlat, nlon=nlon, seed=seed, dt=dt, nsteps=2, over={}, with_albedo=False),
         F=F, k4=k4, **{"ref_" + k: v for k, v in ref.items()},
         **{"init_" + k: st2[k] for k in STATE}, **{"ref_ts_" + k: getattr(m2, k) for k in STATE})


def case_driver_physics(nlat, nlon, seed, nsteps=3):
    """run_simulation.py:1766-1934 + 2063-2146 composed from the REFERENCE's functions
    (physics.*, scripts.run_simulation._advect_scalar_periodic, scipy gaussian_filter), interleaved
    with the reference time_step(Teq, dt) the way the driver calls it (no albedo argument)."""
    from scipy.ndimage import gaussian_filter
    from pygcm import physics as rph
    from pygcm.forcing import ThermalForcing
    from pygcm.orbital import OrbitalSystem
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            from scripts.run_simulation import _advect_scalar_periodic
    finally:
        os.chdir(cwd)
    from qd_oracle.driver import driver_physics_step
    g, mask, alb, fric = surface(nlat, nlon)
    st = perturbed_state((nlat, nlon), seed, cloudy=True, icy=True)
    dt = 300.0
    snaps = {}
    with ref_env({}):
        forcing = ThermalForcing(g, OrbitalSystem())
        m = build_ref_model(g, mask, fric, st)
        for i in range(nsteps):
            t = i * dt
            precip = rph.diagnose_precipitation_hybrid(m, g, D_crit=-1e-7, k_precip=1e5, orog_factor=None,
                                                       smooth_sigma=1.0, beta_div=0.4, renorm=True)
            if np.any(precip > 0):
                P_pos = precip[precip > 0]
                P_ref = float(np.median(P_pos)) if P_pos.size > 0 else 1e-6
            else:
                P_ref = 1e-6
            C_from_P = rph.cloud_from_precip(precip, C_max=0.95, P_ref=P_ref, smooth_sigma=1.0)
            src = rph.parameterize_cloud_cover(m, g, mask)
            tendency = src * (dt / (6 * 3600))
            W_MEM, W_P, W_SRC = 0.4, 0.4, 0.2
            W_sum = W_MEM + W_P + W_SRC
            W_MEM /= W_sum; W_P /= W_sum; W_SRC /= W_sum
            m.cloud_cover = (W_MEM * m.cloud_cover + W_P * C_from_P + W_SRC * np.clip(m.cloud_cover + tendency, 0.0, 1.0))
            m.cloud_cover = np.maximum(m.cloud_cover, np.clip(0.8 * C_from_P, 0.0, 1.0))
            m.cloud_cover = np.clip(m.cloud_cover, 0.0, 1.0)
            cloud_adv = _advect_scalar_periodic(m.cloud_cover, m.u, m.v, dt, g)
            cloud_adv = gaussian_filter(cloud_adv, sigma=0.2, mode="wrap")
            m.cloud_cover = np.clip((1.0 - 0.7) * m.cloud_cover + 0.7 * cloud_adv, 0.0, 1.0)
            insA, insB = forcing.calculate_insolation_components(t)
            m.isr_A, m.isr_B = insA, insB
            m.isr = insA + insB
            ice_frac = 1.0 - np.exp(-np.maximum(m.h_ice, 0.0) / 0.5)
            cloud_for_rad = getattr(m, "cloud_eff_last", m.cloud_cover)
            albedo = rph.calculate_dynamic_albedo(cloud_for_rad, m.T_s, alb.copy(), 0.6, 0.5, land_mask=mask, ice_frac=ice_frac)
            Teq = forcing.calculate_equilibrium_temp(t, albedo)
            m.time_step(Teq, dt)
            if i == 0:
                snaps.update(s1_precip=precip.copy(), s1_albedo=albedo.copy(), s1_C_from_P=C_from_P.copy(), s1_src=src.copy())
    P = qo.defaults()
    og, om = build_oracle_model(g, mask, fric, P, st)
    of = qo.Forcing(og)
    for i in range(nsteps):
        t = i * dt
        precip_o, albedo_o = driver_physics_step(om, og, P, alb, mask, dt)
        a_, b_ = of.insolation_components(t)
        om.isr_A, om.isr_B, om.isr = a_, b_, a_ + b_
        om.time_step(of.equilibrium_temp(t, albedo_o), dt)
    for k in STATE:
        print(f"    {k:18s} oracle-vs-ref maxrel {maxrel(getattr(om, k), getattr(m, k)):.2e}")
    print(f"    precip(last)       oracle-vs-ref maxrel {maxrel(precip_o, precip):.2e}")
    print(f"    albedo(last)       oracle-vs-ref maxrel {maxrel(albedo_o, albedo):.2e}")
    save(f"driverphys_{nlat}x{nlon}", dict(kind="driver_physics", nlat=nlat, nlon=nlon, seed=seed, nsteps=nsteps, dt=dt),
         **{"init_" + k: st[k] for k in STATE}, **{"ref_" + k: getattr(m, k) for k in STATE},
         ref_precip_last=precip, ref_albedo_last=albedo, **snaps)


def main():
    only = sys.argv[1:]

    def want(n):
        return (not only) or any(o in n for o in only)
    if want("ops"):
        for (a, b, s) in ((19, 36, 1), (37, 72, 2)):
            print(f"[operators {a}x{b}]")
            quiet_case = case_operators(a, b, s)
    ts_cases = [
        # name, nlat, nlon, nsteps, overrides, with_albedo, seed, wet, cloudy, icy
        ("ts_19x36_default_noalb", 19, 36, 12, {}, False, None, False, False, False),
        ("ts_19x36_default_alb", 19, 36, 12, {}, True, None, False, False, False),
        ("ts_19x36_energy_primitive", 19, 36, 12, {"energy_w": 1.0, "mom_scheme": 1}, True, None, False, False, False),
        ("ts_37x72_energy_wet_cloudy_icy", 37, 72, 8, {"energy_w": 1.0}, True, 5, True, True, True),
        ("ts_37x72_perturbed_noalb", 37, 72, 7, {"filter_type": "hyper4"}, False, 6, False, True, False),
        ("ts_37x72_lwv1_noseaice", 37, 72, 6, {"energy_w": 0.7, "lw_v2": 0, "seaice_enabled": 0, "gh_lock": 0}, True, 7, False, True, True),
        ("ts_19x36_spectral_diffq", 19, 36, 6, {"spec_every": 2, "diff_q": 1, "diff_cloud": 1, "k4_nsub": 2, "shapiro_every": 3}, True, 8, False, True, False),
        ("ts_19x36_pcondref_scalar_k4", 19, 36, 4, {"energy_w": 1.0, "pcond_ref": 2e-6, "k4_u": 1.0e14, "k4_q": 0.0, "tau_cond": 900.0}, True, 9, True, True, True),
    ]
    for c in ts_cases:
        if want(c[0]):
            print(f"[{c[0]}]")
            quiet(lambda: None)
            case_time_step(c[0], c[1], c[2], c[3], c[4], c[5], seed=c[6], wet=c[7], cloudy=c[8], icy=c[9])
    oc_cases = [
        ("ocean_19x36_default", 19, 36, 4, {}, 21, False),
        ("ocean_37x72_strong", 37, 72, 3, {}, 22, True),
        ("ocean_19x36_clamp_shapiro", 19, 36, 8, {"ocean_outlier": "clamp", "ocean_shapiro_n": 1}, 23, True),
        ("ocean_37x72_nsub", 37, 72, 3, {"ocean_cfl": 0.02}, 24, True),
        ("ocean_19x36_nanpoison", 19, 36, 3, {}, 25, False, True),
    ]
    for c in oc_cases:
        if want(c[0]):
            print(f"[{c[0]}]")
            case_ocean(*c)
    if want("driverphys"):
        for (a, b, sd) in ((19, 36, 41), (37, 72, 42)):
            print(f"[driver physics {a}x{b}]")
            case_driver_physics(a, b, sd)
    if want("nonfinite"):
        print("[non-finite inputs 19x36]")
        case_nonfinite(19, 36, 71)
    if want("bands"):
        print("[spectral bands 19x36]")
        case_spectral_bands(19, 36)
    if want("eco"):
        print("[ecology canopy + individuals 19x36]")
        case_ecology(19, 36, 81)
        case_ecology(19, 36, 82, variant="rayleigh")
    if want("phyto"):
        for (a, b, s) in ((19, 36, 61), (37, 72, 62)):
            print(f"[phyto transport {a}x{b}]")
            case_phyto(a, b, s)
    if want("orog"):
        for (a, b, s) in ((19, 36, 51), (37, 72, 52)):
            print(f"[orography {a}x{b}]")
            case_orography(a, b, s)
    if want("physics"):
        for (a, b, s) in ((19, 36, 31), (37, 72, 32)):
            print(f"[physics {a}x{b}]")
            case_physics(a, b, s)


if __name__ == "__main__":
    main()
