"""GPU (-m gpu): the per-physics-step ecology (BASELINE config 5; SURVEY.md 8(f)3 stages 2-4) through the C-ABI, against the
fixture the reference's own classes produced (tests/golden/eco_19x36.npz) and against the oracle inside the whole driver loop.

Tolerances: the alpha maps and the canopy factor go through one device exp (<= 1 ulp of f, values in [0, 1]): 1e-15 absolute.
E_day is a plain sum of products: exact.  Individuals: the band split and a 16-term dot product whose association differs from
einsum's: 1e-14 relative.
"""
import numpy as np
import pytest

from util import load_golden, relerr

pytestmark = pytest.mark.gpu

ECO_ENV = {"QD_ECO_NS": "4", "QD_ECO_COHORT_K": "2", "QD_ECO_LIGHT_UPDATE_EVERY_HOURS": "0.5", "QD_ECO_SUBSTEP_EVERY_NPHYS": "2",
           "QD_ECO_LAI_K": "0.6", "QD_ECO_SOIL_REFLECT": "0.18", "QD_ECO_INDIV_SAMPLE_FRAC": "0.3", "QD_ECO_INDIV_PER_CELL": "5",
           "QD_ECO_INDIV_SUBSTEPS_PER_DAY": "10"}


def _setenv(monkeypatch, extra=None):
    import os
    for k in list(os.environ):
        if k.startswith("QD_ECO_") or k.startswith("QD_PHYTO_"):
            monkeypatch.delenv(k)
    full = extra is not None and "QD_ECO_NS" in extra                # a fixture's complete environment replaces the default one
    for k, v in (extra if full else {**ECO_ENV, **(extra or {})}).items():
        monkeypatch.setenv(k, v)


def _device(nlat, nlon, land_mask):
    import qingdai_amd as qa
    from qingdai_amd.device import Device
    p = qa.QdParams(); p.has_csmap = 0
    dev = Device(qa.SphericalGrid(nlat, nlon), p)
    dev.upload_now("LAND_MASK", land_mask)
    return dev


ECO_CASES = ["eco_19x36", "eco_19x36_rayleigh"]       # second fixture: Rayleigh band weights, 8 bands, per-species genes from the env


@pytest.mark.parametrize("f32_maps", [False, True])
@pytest.mark.parametrize("case", ECO_CASES)
def test_canopy_alpha_sequence_vs_reference(gpu, monkeypatch, case, f32_maps):
    """EcologyAdapter.step_subdaily over PopulationManager (adapter.py:140-186, population.py:252-294,895-915): ten steps with
    the LAI stack replaced twice, so that the first-call, LAI-change and clock triggers of the canopy cache all fire; alpha on
    every second step (QD_ECO_SUBSTEP_EVERY_NPHYS=2).

    f32_maps (QD_ECO_F32, BASELINE configs[4] "f32 mixed precision"): LAI_tot, its snapshot, the canopy factor and the alpha maps are
    STORED as f32 on the device; arithmetic, the LAI plane sum, the lai-delta reduction and E_day stay f64.  Same trigger sequence,
    E_day bit-identical, maps within a few f32 roundings (2^-24 = 6e-8 each: LAI -> f -> alpha) of the reference's f64 values --
    the measured deviation is printed and bounded at 5e-7."""
    from qingdai_amd.ecology import EcologyAdapter
    meta, d = load_golden(case)
    _setenv(monkeypatch, meta["env"])
    dev = _device(meta["nlat"], meta["nlon"], d["land_mask"])
    eco = EcologyAdapter(dev.grid, d["land_mask"], dev=dev, albedo_couple=True, f32_maps=f32_maps)
    tol = 5e-7 if f32_maps else 1e-15
    worst = 0.0
    assert eco.alpha_leaf_scalar == meta["leaf_scalar"] and eco.pop.LAI_layers_SK.shape == d["L0"].shape
    assert eco.bands.nbands == meta["nb"]
    eco.pop.push_layers(d["L0"], init=True)
    if f32_maps:                                                                       # f64 plane sum, one rounding to f32
        assert np.array_equal(eco.pop.total_LAI(), np.sum(d["L0"], axis=(0, 1)).astype(np.float32).astype(np.float64))
    else:
        assert np.array_equal(eco.pop.total_LAI(), np.sum(d["L0"], axis=(0, 1)))      # plane-by-plane sum is numpy's order
    for i, st in enumerate(meta["steps"]):
        if i == 2:
            eco.pop.push_layers(d["L1"])
        if i == 5:
            eco.pop.push_layers(d["L2"])
        a = eco.step_subdaily(d[f"insA_{i}"] + d[f"insB_{i}"], 0.3, meta["dt"])
        assert (a is not None) == st["returned"], i
        assert eco.pop.state()["n_recompute"] == st["n_recompute"], (i, eco.pop.state())
        if a is not None:
            ref = d[f"ref_alpha_{i}"]
            assert np.array_equal(np.isnan(a), np.isnan(ref)), i
            worst = max(worst, float(np.nanmax(np.abs(a - ref))))
            assert np.nanmax(np.abs(a - ref)) < tol, (i, np.nanmax(np.abs(a - ref)))
    assert np.array_equal(eco.pop.E_day, d["ref_E_day"])
    dev._host.pop("ECO_F", None)
    worst = max(worst, float(np.max(np.abs(dev.get("ECO_F") - d["ref_f_cached"]))))
    assert np.max(np.abs(dev.get("ECO_F") - d["ref_f_cached"])) < tol
    # daily banded alpha (population.py:875-893 + run_simulation.py:1843-1844), species reflectance from the default genes
    assert np.array_equal(eco.pop._species_R_leaf, d["R_species"]) and np.array_equal(eco.pop.species_weights, d["species_w"])
    got = eco.banded_alpha()
    worst = max(worst, float(np.max(np.abs(got - d["ref_alpha_banded"]))))
    assert np.max(np.abs(got - d["ref_alpha_banded"])) < tol
    A, w_b = eco.get_surface_albedo_bands()
    assert np.array_equal(np.isnan(A), np.isnan(d["ref_A_bands"])) and np.nanmax(np.abs(A - d["ref_A_bands"])) < tol
    assert np.array_equal(w_b, d["ref_w_b"])
    print(f"{case} f32_maps={f32_maps}: largest deviation of alpha / f / banded alpha from the reference {worst:.3e}")
    assert (worst > 0.0) == f32_maps or not f32_maps
    dev.close()


def test_adapter_without_population_vs_reference(gpu, monkeypatch):
    """QD_ECO_USE_LAI=0, the adapter's M1 branch (adapter.py:79-80,162-166): no population, no E_day, the scalar leaf alpha on
    land on every 2nd call; a pool of individuals cannot be built on it (individuals.py:67-69)."""
    from qingdai_amd.ecology import EcologyAdapter, IndividualPool
    _setenv(monkeypatch, {"QD_ECO_USE_LAI": "0"})
    meta, d = load_golden("eco_19x36")
    dev = _device(meta["nlat"], meta["nlon"], d["land_mask"])
    eco = EcologyAdapter(dev.grid, d["land_mask"], dev=dev, albedo_couple=True)
    assert eco.pop is None
    got = [eco.step_subdaily(d["insA_0"] + d["insB_0"], 0.3, meta["dt"]) for _ in range(4)]
    assert [a is not None for a in got] == [False, True, False, True]
    assert np.array_equal(got[1], d["ref_alpha_m1"], equal_nan=True) and np.array_equal(got[3], d["ref_alpha_m1"], equal_nan=True)
    dev._host.pop("ECO_EDAY", None)
    assert np.all(dev.get("ECO_EDAY") == 0.0)
    with pytest.raises(RuntimeError, match="requires EcologyAdapter.pop"):
        IndividualPool(dev.grid, d["land_mask"], eco)
    dev.close()


@pytest.mark.parametrize("case", ECO_CASES)
def test_individual_pool_vs_reference(gpu, monkeypatch, case):
    """IndividualPool (individuals.py:37-191): the mirror draws the reference's pool (same cells, species, jitter), and 30 long
    physics steps fire the 12 sub-steps the reference fired, with its energies and stress days."""
    import qd_oracle as qo
    from qingdai_amd.ecology import EcologyAdapter, IndividualPool
    meta, d = load_golden(case)
    _setenv(monkeypatch, meta["env"])
    nlat, nlon = meta["nlat"], meta["nlon"]
    dev = _device(nlat, nlon, d["land_mask"])
    eco = EcologyAdapter(dev.grid, d["land_mask"], dev=dev, albedo_couple=True)
    pool = IndividualPool(dev.grid, d["land_mask"], eco, day_seconds=meta["ind_day"], soil_cap=1.0)
    assert np.array_equal(pool.sample_j, d["ind_sample_j"]) and np.array_equal(pool.sample_i, d["ind_sample_i"])
    assert np.array_equal(pool.indiv_cell_index, d["ind_cell"]) and np.array_equal(pool.indiv_tol, d["ind_tol"])
    assert np.array_equal(pool.indiv_Ab, d["ind_Ab"])
    of = qo.Forcing(qo.Grid(nlat, nlon))
    fired = []
    for i in range(30):
        a_, b_ = of.insolation_components(i * meta["ind_dt"])
        if pool.try_substep(a_, b_, eco, d["ind_soil"] if i == 0 else None, meta["ind_dt"], meta["ind_day"]):
            fired.append(i)
    assert fired == meta["ind_fired"]
    E, S = pool.indiv_E_day, pool.indiv_water_stress_days
    print("individuals", relerr(E, d["ref_ind_E_day"]), relerr(S, d["ref_ind_stress"]))
    assert relerr(E, d["ref_ind_E_day"]) < 1e-14 and np.array_equal(S, d["ref_ind_stress"])
    pool.reset()
    assert np.all(pool.indiv_E_day == 0.0)
    # f32 storage of the coefficient table (QD_ECO_F32 / BASELINE configs[4] "f32 mixed"): same sub-steps, f32 rounding of Ab only
    pool32 = IndividualPool(dev.grid, d["land_mask"], eco, day_seconds=meta["ind_day"], soil_cap=1.0, f32_storage=True)
    for i in range(30):
        a_, b_ = of.insolation_components(i * meta["ind_dt"])
        pool32.try_substep(a_, b_, eco, None, meta["ind_dt"], meta["ind_day"])
    e32 = relerr(pool32.indiv_E_day, d["ref_ind_E_day"])
    print("f32 table:", e32)
    assert 0 < e32 < 1e-7 and np.array_equal(pool32.indiv_water_stress_days, d["ref_ind_stress"])
    # a cell index outside the grid never reaches the device
    from qingdai_amd._lib import QdError
    pool.sample_j = pool.sample_j.copy(); pool.sample_j[0] = nlat
    with pytest.raises(QdError, match="outside the grid"):
        pool.configure()
    dev.close()


@pytest.mark.parametrize("variant", ["lai", "bands_water"])
def test_driver_loop_with_ecology_vs_oracle(gpu, monkeypatch, variant):
    """The whole driver iteration with the ecology inside the resident loop (qd_step_n bit5) against DriverOracle carrying the
    oracle's EcoCoupling + IndividualSubstep: the alpha blend sits between the glacier mask and the snow blend
    (run_simulation.py:2075-2141), E_day rides on the forcing kernel, the individuals read W_land before the bucket update.
    `lai`: W_LAI = 0.8, alpha every 2nd step, LAI stack replaced after 3 steps.  `bands_water`: banded land alpha and an
    ocean-colour map on top (QD_ECO_BANDS_COUPLE, phytoplankton coupling)."""
    import qd_oracle as qo
    from qd_oracle import ecology as oeco, spectral as osp
    from qd_oracle.driver import DriverOracle
    from qingdai_amd.driver import Simulation
    extra = {"QD_ECO_LAI_ALBEDO_WEIGHT": "0.8", "QD_ECO_INDIV_SAMPLE_FRAC": "0.05", "QD_ECO_INDIV_PER_CELL": "7",
             "QD_ECO_INDIV_SUBSTEPS_PER_DAY": "120", "QD_ECO_LIGHT_UPDATE_EVERY_HOURS": "0.25"}
    if variant == "bands_water":
        extra.update({"QD_ECO_BANDS_COUPLE": "1", "QD_PHYTO_ENABLE": "1", "QD_PHYTO_ALBEDO_COUPLE": "1"})
    _setenv(monkeypatch, extra)
    nlat, nlon, nsteps = 61, 96, 6
    sim = Simulation(nlat, nlon, params=__import__("qingdai_amd").QdParams(), use_ocean=True, quiet=True)
    assert sim.eco is not None and sim.indiv is not None
    r = np.random.default_rng(5)
    lat = np.deg2rad(sim.grid.lat_mesh)
    land = (sim.land_mask == 1)
    h0 = 8000.0 - 10500.0 * np.sin(lat) ** 2
    Ts0 = 262.0 + 36.0 * np.cos(lat) ** 2
    S0 = np.where(land & (np.abs(sim.grid.lat_mesh) > 55), 30.0, 0.0)
    W0 = np.where(land, 40.0 * r.random((nlat, nlon)), 0.0)
    sim.gcm.h, sim.gcm.T_s = h0, Ts0
    sim.dev.set("S_SNOW", S0); sim.dev.set("W_LAND", W0)
    S, K = sim.eco.pop.LAI_layers_SK.shape[:2]
    L0 = np.abs(r.normal(0.5, 0.4, (S, K, nlat, nlon))) * land
    L1 = L0 * (1.0 + 0.5 * r.random(L0.shape))
    sim.eco.pop.push_layers(L0, init=True)
    # oracle twin
    g, P = qo.Grid(nlat, nlon), qo.defaults()
    m = qo.AtmosOracle(g, sim.friction, sim.land_mask, P, C_s_map=np.where(land, 3e6, P.Cs_ocean).astype(float))
    m.h, m.T_s = h0.copy(), Ts0.copy()
    oc = qo.OceanOracle(g, sim.land_mask, P, init_Ts=np.full((nlat, nlon), 288.0))
    drv = DriverOracle(g, m, oc, qo.Forcing(g), sim.land_mask, sim.base_albedo, P)
    drv.S_snow, drv.W_land = S0.copy(), W0.copy()
    ob = osp.make_bands(16, 380.0, 780.0)
    opop = oeco.CanopyPopulation(sim.land_mask, L0, k_canopy=0.6, light_update_every_hours=0.25, recompute_lai_delta=0.05)
    oad = oeco.EcoAdapter(opop, oeco.leaf_scalar(ob), soil_ref=0.18, substep_every_nphys=2)
    drv.eco = oeco.EcoCoupling(oad, w_lai=0.8)
    if variant == "bands_water":
        R_eff = oeco.effective_leaf_reflectance(sim.eco.pop.species_weights, sim.eco.pop._species_R_leaf)
        # get_surface_albedo_bands builds the canopy cache on demand (population.py:837-838); it then survives into the loop
        drv.eco.alpha_banded = oeco.banded_alpha(opop.surface_albedo_bands(R_eff, 0.18), oeco.band_weights(ob))
        water = np.where(~land, 0.05 + 0.1 * r.random((nlat, nlon)), np.nan)
        water[0, :] = np.nan                                       # non-finite ocean cells keep their base albedo
        drv.eco.ocean_alpha = water
        sim.dev.upload_now("WATER_ALPHA", water)
        got_b = sim.eco.banded_alpha()
        assert np.max(np.abs(got_b - drv.eco.alpha_banded)) < 1e-15
    pool = sim.indiv
    drv.indiv = oeco.IndividualSubstep(pool.sample_j, pool.sample_i, pool.indiv_cell_index, pool.indiv_Ab, pool.indiv_tol, 120)
    drv.indiv_bands, drv.indiv_day, drv.soil_cap = ob, sim.day_seconds, 50.0
    sim.run_steps(3)
    sim.eco.pop.push_layers(L1)
    sim.run_steps(nsteps - 3)
    for i in range(nsteps):
        if i == 3:
            opop.layers = L1.copy()
        drv.step(i * 300.0, 300)
    st = sim.eco.pop.state()
    print(variant, st, "oracle recomputes", opop.n_recompute, "individual sub-steps", drv.indiv.n_fired)
    assert st["step_count"] == nsteps and drv.indiv.n_fired >= 2
    pairs = {"u": (sim.gcm.u, m.u), "h": (sim.gcm.h, m.h), "T_s": (sim.gcm.T_s, m.T_s), "q": (sim.gcm.q, m.q),
             "cloud": (sim.gcm.cloud_cover, m.cloud_cover), "albedo": (sim.dev.get("ALBEDO"), drv.albedo),
             "S_snow": (sim.dev.get("S_SNOW"), drv.S_snow), "W_land": (sim.dev.get("W_LAND"), drv.W_land),
             "E_day": (sim.eco.pop.E_day, opop.E_day), "SST": (sim.ocean.Ts, oc.Ts),
             "indiv_E": (pool.indiv_E_day, drv.indiv.E_day), "indiv_stress": (pool.indiv_water_stress_days, drv.indiv.stress_days)}
    errs = {k: relerr(a, b) for k, (a, b) in pairs.items()}
    print(errs)
    for k, e in errs.items():
        assert e < 1e-9, (k, e)
    # the blend really acted: the albedo over snow-free, ice-free land differs from the run without ecology
    a_noeco = np.clip(sim.base_albedo, 0, 1)
    assert np.abs(drv.eco.last_alpha[land] - a_noeco[land]).max() > 0.01
    sim.dev.close()


def test_daily_hook_at_day_boundaries(gpu, monkeypatch):
    """The seam to the host-side daily ecology (run_simulation.py:1784-1864): `Simulation(daily_hook=...)` is called inside the
    step that completes a planet-day, before the rest of that step, with the soil index of the previous step's W_land (zero on
    ice sheets); it reads E_day, resets it and pushes new LAI layers.  A shortened day (5 steps) and a toy growth rule, mirrored
    on DriverOracle."""
    import qd_oracle as qo
    from qd_oracle import ecology as oeco, spectral as osp
    from qd_oracle.driver import DriverOracle
    from qingdai_amd.driver import Simulation
    _setenv(monkeypatch, {"QD_ECO_SUBSTEP_EVERY_NPHYS": "1", "QD_ECO_LIGHT_UPDATE_EVERY_HOURS": "6"})
    nlat, nlon, nsteps, day = 61, 96, 12, 1500.0
    calls = []

    def grow(L, E, soil):
        return L * (1.0 + 2.0e-6 * E)[None, None] * (0.6 + 0.4 * soil)[None, None]

    def hook(sim_, soil_idx, glacier):
        E = sim_.eco.pop.E_day
        calls.append((sim_._step_index, float(E.sum()), float(soil_idx.sum()), int(glacier.sum())))
        sim_.eco.pop.E_day = 0.0
        sim_.eco.pop.push_layers(grow(sim_.eco.pop.LAI_layers_SK, E, soil_idx))

    sim = Simulation(nlat, nlon, params=__import__("qingdai_amd").QdParams(), use_ocean=False, quiet=True, individuals=False,
                     daily_hook=hook)
    sim.day_seconds = day
    r = np.random.default_rng(8)
    lat = np.deg2rad(sim.grid.lat_mesh)
    land = (sim.land_mask == 1)
    h0 = 8000.0 - 10500.0 * np.sin(lat) ** 2
    Ts0 = 262.0 + 36.0 * np.cos(lat) ** 2
    S0 = np.where(land & (np.abs(sim.grid.lat_mesh) > 55), 60.0, 0.0)             # >= 50 mm: ice sheet from the first step
    W0 = np.where(land, 40.0 * r.random((nlat, nlon)), 0.0)
    sim.gcm.h, sim.gcm.T_s = h0, Ts0
    sim.dev.set("S_SNOW", S0); sim.dev.set("W_LAND", W0)
    S, K = sim.eco.pop.LAI_layers_SK.shape[:2]
    L0 = np.abs(r.normal(0.5, 0.4, (S, K, nlat, nlon))) * land
    sim.eco.pop.push_layers(L0, init=True)
    g, P = qo.Grid(nlat, nlon), qo.defaults()
    m = qo.AtmosOracle(g, sim.friction, sim.land_mask, P, C_s_map=np.where(land, 3e6, P.Cs_ocean).astype(float))
    m.h, m.T_s = h0.copy(), Ts0.copy()
    drv = DriverOracle(g, m, None, qo.Forcing(g), sim.land_mask, sim.base_albedo, P)
    drv.S_snow, drv.W_land = S0.copy(), W0.copy()
    ob = osp.make_bands(16, 380.0, 780.0)
    opop = oeco.CanopyPopulation(sim.land_mask, L0, k_canopy=0.6, light_update_every_hours=6.0, recompute_lai_delta=0.05)
    drv.eco = oeco.EcoCoupling(oeco.EcoAdapter(opop, oeco.leaf_scalar(ob), soil_ref=0.18, substep_every_nphys=1), w_lai=1.0)
    sim.run_steps(7)
    sim.run_steps(nsteps - 7)
    accum, ocalls = 0.0, []
    for i in range(nsteps):
        accum += 300.0
        while accum >= day:
            accum -= day
            soil = np.clip(drv.W_land / 50.0, 0.0, 1.0) * (~drv.glacier)
            ocalls.append((i, float(opop.E_day.sum()), float(soil.sum()), int(drv.glacier.sum())))
            opop.layers = grow(opop.layers, opop.E_day, soil)
            opop.E_day = np.zeros_like(opop.E_day)
        drv.step(i * 300.0, 300)
    print(calls, ocalls, sim.eco.pop.state(), opop.n_recompute)
    assert [c[0] for c in calls] == [4, 9] == [c[0] for c in ocalls]
    for a, b in zip(calls, ocalls):
        assert abs(a[1] - b[1]) <= 1e-12 * abs(b[1]) and abs(a[2] - b[2]) <= 1e-9 * abs(b[2]) and a[3] == b[3] and b[3] > 0
    assert sim.eco.pop.state()["n_recompute"] == opop.n_recompute == 3           # first step + the two LAI jumps
    assert relerr(sim.eco.pop.total_LAI(), opop.total_LAI()) < 1e-12
    assert relerr(sim.eco.pop.E_day, opop.E_day) < 1e-14
    assert relerr(sim.dev.get("ALBEDO"), drv.albedo) < 1e-9 and relerr(sim.gcm.T_s, m.T_s) < 1e-9
    sim.dev.close()
