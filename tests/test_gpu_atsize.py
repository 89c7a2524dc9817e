"""GPU (-m gpu): oracle parity AT THE SIZES THE BENCHMARK AND THE REFERENCE DRIVER USE.

* BASELINE configs[2] (what bench.py times): 721 x 1440, the loop forcing -> driver physics -> time_step(Teq, dt, albedo) with
  QD_ENERGY_W=1 -> ocean coupling, three steps, device against qd_oracle.  This is the check that would catch a size-dependent
  indexing error in exactly the kernels the bench times (strip / tile dealing at 25 x 30 strips, n_sub = 11-12 ocean sub-steps,
  pole strips, windowed medians).
* The reference's own driver run (scripts.run_simulation.main() at its native 121 x 240, 24 steps):
  the DEVICE driver against the reference's known-answer sums.
* BASELINE configs[3]: 1441 x 2880 in 8 latitude bands against the whole-globe handle.
"""
import numpy as np
import pytest

from util import relerr
from test_oracle_polar_noise_cpu import POLAR_ROWS, atsize_wind

pytestmark = pytest.mark.gpu

# Relative to each field's max-norm (whole-run bound of SURVEY.md 8(d) for the atmosphere).
ATM_TOL, OCN_TOL = 1e-9, 1e-7


def _build(with_ocean):
    import qd_oracle as qo
    from qd_oracle.driver import DriverOracle
    import qingdai_amd as qa
    from qingdai_amd.topography import create_land_sea_mask, generate_base_properties
    nlat, nlon = 721, 1440
    over = dict(energy_w=1.0, cloud_couple=1)
    grid = qa.SphericalGrid(nlat, nlon)
    mask = create_land_sea_mask(grid)
    base_albedo, friction = generate_base_properties(mask)
    Cs_ocean = 1000.0 * 4200.0 * 50.0
    csmap = np.where(mask == 1, 3e6, Cs_ocean).astype(float)
    # ---- device: exactly bench.py's build_case
    m = qa.SpectralModel(grid, friction, H=8000, tau_rad=10 * 24 * 3600, greenhouse_factor=0.40, C_s_map=csmap, land_mask=mask,
                         Cs_ocean=Cs_ocean, Cs_land=3e6, Cs_ice=5e6, params=qa.QdParams(**over))
    oc = qa.WindDrivenSlabOcean(grid, mask, 50.0, init_Ts=np.full((nlat, nlon), 288.0)) if with_ocean else None
    # a spun-up wind field (the bench reaches it after its warm-up steps): it puts the slab ocean at the 12-13 sub-steps per step
    # the benchmark runs with; a cold start would only see n_sub = 2-5.  Shared with the CPU test that measures the polar noise.
    lat = np.deg2rad(grid.lat_mesh); lon = np.deg2rad(grid.lon_mesh)
    u0, v0 = atsize_wind(lat, lon)
    m.u, m.v = u0, v0
    m._dev.upload_now("BASE_ALBEDO", base_albedo)
    forcing = qa.ThermalForcing(grid, qa.OrbitalSystem())
    # ---- oracle: the same composition
    g = qo.Grid(nlat, nlon)
    P = qo.defaults(**over)
    om = qo.AtmosOracle(g, friction, mask, P, C_s_map=csmap)
    om.u, om.v = u0.copy(), v0.copy()
    oo = qo.OceanOracle(g, mask, P, init_Ts=np.full((nlat, nlon), 288.0)) if with_ocean else None
    d = DriverOracle(g, om, oo, qo.Forcing(g), mask, base_albedo, P)
    return m, oc, forcing, om, oo, d


def test_config2_atmosphere_and_driver_physics_vs_oracle_at_721x1440(gpu):
    """forcing -> driver physics (hybrid precipitation, clouds, albedo; medians over 1 M cells) -> time_step(Teq, dt, albedo) with
    QD_ENERGY_W=1, three steps, every cell of the 721 x 1440 grid (25 x 30 strips of the fused kernel incl. both pole strips)."""
    m, _, forcing, om, _, d = _build(False)
    nsteps, dt = 3, 300.0
    m._dev.step_n(forcing.star_table([i * dt for i in range(nsteps)]), dt, with_ocean=False, with_physics=True, pass_albedo=True)
    for i in range(nsteps):
        d.step(i * dt, dt, pass_albedo=True, commit=False)
    pairs = {"u": (m.u, om.u), "v": (m.v, om.v), "h": (m.h, om.h), "T_s": (m.T_s, om.T_s), "q": (m.q, om.q),
             "cloud": (m.cloud_cover, om.cloud_cover), "h_ice": (m.h_ice, om.h_ice), "albedo": (m._dev.get("ALBEDO"), d.albedo),
             "precip": (m._dev.get("PRECIP"), d.precip)}
    errs = {k: relerr(a, b) for k, (a, b) in pairs.items()}
    print(errs)
    for k, e in errs.items():
        assert e < ATM_TOL, (k, e)


def test_config2_coupled_loop_vs_oracle_at_721x1440(gpu):
    """The same loop with the slab ocean coupled in (13 sub-steps per step: k_ocn_stream + k_ocn_tail_stream), two steps.

    What can be compared: at this resolution the two polar ocean rows are violently unstable IN THE REFERENCE ARITHMETIC ITSELF --
    eta sits at its +-5 m clip and flips sign there.  tests/test_oracle_polar_noise_cpu.py measures it (the oracle against the oracle
    with the wind perturbed by 1e-15): after ONE coupled step the two runs differ by O(1) on rows 0-7 and 713-720 and agree to
    2e-10 (currents) / 3e-14 (atmosphere) everywhere else.  So after the first step every row but POLAR_ROWS = 16 next to each pole
    is compared -- that includes the second pole-side strip of k_ocn_stream and every strip of the tail kernel but the first two --
    and after the second step every row but 48 next to each pole (the noise travels at most ~8 rows per step through the stencils
    that read it).  The pole strips themselves are compared with the oracle by test_ocean_single_substep_every_row_vs_oracle_at_
    721x1440 below, before the instability has anything to amplify.  The sub-step counts must agree exactly, and the SST written
    back into T_s couples the two models everywhere inside the window.  eta gets a wider bound: every sub-step subtracts the
    area-weighted GLOBAL mean of eta, which carries the polar rows' O(1) noise (at cos-latitude weight) into every cell as a
    uniform shift of ~1e-7 of the clip value."""
    m, oc, forcing, om, oo, d = _build(True)
    nsteps, dt = 2, 300.0
    stars = forcing.star_table([i * dt for i in range(nsteps)])
    nsubs = []
    n = 721

    def compare(rows, what):
        pairs = {"u": (m.u, om.u), "v": (m.v, om.v), "h": (m.h, om.h), "T_s": (m.T_s, om.T_s), "q": (m.q, om.q),
                 "cloud": (m.cloud_cover, om.cloud_cover), "uo": (oc.uo, oo.uo), "vo": (oc.vo, oo.vo), "eta": (oc.eta, oo.eta),
                 "SST": (oc.Ts, oo.Ts)}
        errs = {k: relerr(a[rows], b[rows]) for k, (a, b) in pairs.items()}
        print(what, "n_sub", nsubs, errs)
        for k, e in errs.items():
            assert e < (1e-6 if k == "eta" else OCN_TOL if k in ("uo", "vo") else ATM_TOL), (what, k, e)

    m._dev.step_n(stars[:1], dt, with_ocean=True, with_physics=True, pass_albedo=True)
    d.step(0.0, dt, pass_albedo=True, commit=False)
    nsubs.append(oo.last_n_sub)
    assert m._dev.last_ocean_nsub() == nsubs[-1]
    compare(slice(POLAR_ROWS, n - POLAR_ROWS), f"after step 1, rows {POLAR_ROWS} .. {n - POLAR_ROWS - 1}:")
    m._dev.step_n(stars[1:2], dt, with_ocean=True, with_physics=True, pass_albedo=True)
    d.step(dt, dt, pass_albedo=True, commit=False)
    nsubs.append(oo.last_n_sub)
    assert m._dev.last_ocean_nsub() == nsubs[-1] and min(nsubs) >= 12, nsubs      # the sub-step counts the bench runs with
    compare(slice(48, n - 48), f"after step 2, rows 48 .. {n - 49}:")


def test_ocean_single_substep_every_row_vs_oracle_at_721x1440(gpu):
    """The pole-side strips of k_ocn_stream / k_ocn_tail_stream at the benchmark size, compared with the oracle on EVERY row: one
    ocean step of 20 s (one sub-step: nothing has been amplified yet) from a smooth, fast state -- currents up to the 3 m/s cap so
    that the outlier filter and its np.roll neighbours across the poles are exercised, eta inside its clip, a warm-pool SST, wind
    stress from a 250 m/s jet, Q_net heating with an ice mask -- then a second step that applies the deferred "eta - mean, clip" of
    the first on load.  Bounds: 1e-12 of each field's max-norm, 1e-10 for eta (operators agree to rounding; one step has no time to grow it)."""
    import qd_oracle as qo
    import qingdai_amd as qa
    from qingdai_amd.topography import create_land_sea_mask
    nlat, nlon = 721, 1440
    grid = qa.SphericalGrid(nlat, nlon)
    mask = create_land_sea_mask(grid)
    ocean = mask == 0
    lat = np.deg2rad(grid.lat_mesh); lon = np.deg2rad(grid.lon_mesh)
    r = np.random.default_rng(5)
    uo = (2.6 * np.cos(lat) ** 0.25 * np.sin(2 * lon + lat) + 0.6 * r.standard_normal((nlat, nlon))) * ocean
    vo = (2.2 * np.cos(3 * lon) * np.cos(lat) ** 0.25 + 0.6 * r.standard_normal((nlat, nlon))) * ocean
    uo[[0, 1, -2, -1]] = 4.0 * ocean[[0, 1, -2, -1]]        # fast pole rows: mean4 reads row -1 as row n-1 and row n as row 0 (np.roll)
    vo[[0, -1]] = -3.0 * ocean[[0, -1]]
    eta = (3.0 * np.sin(3 * lat) * np.cos(2 * lon) + 0.5 * r.standard_normal((nlat, nlon))) * ocean
    sst = 288.0 + 12.0 * np.cos(lat) ** 2 + 1.5 * np.sin(4 * lon)
    u_atm = 250.0 * np.cos(lat) * (1.0 + 0.1 * np.sin(3 * lon)); v_atm = 60.0 * np.sin(2 * lat) * np.cos(2 * lon)
    qnet = 150.0 * np.cos(lat) - 60.0 + 20.0 * r.standard_normal((nlat, nlon))
    ice = (np.abs(np.rad2deg(lat)) > 72.0) & ocean
    P = qo.defaults()
    oo = qo.OceanOracle(qo.Grid(nlat, nlon), mask, P, init_Ts=sst.copy())
    oo.uo, oo.vo, oo.eta = uo.copy(), vo.copy(), eta.copy()
    oc = qa.WindDrivenSlabOcean(grid, mask, 50.0, init_Ts=sst)
    oc.uo, oc.vo, oc.eta = uo, vo, eta
    for k in range(2):
        oc.step(20.0, u_atm, v_atm, Q_net=qnet, ice_mask=ice)
        oo.step(20.0, u_atm, v_atm, Q_net=qnet, ice_mask=ice)
        assert oc.last_n_sub == oo.last_n_sub == 1
        errs = {kk: relerr(a, b) for kk, (a, b) in {"uo": (oc.uo, oo.uo), "vo": (oc.vo, oo.vo), "eta": (oc.eta, oo.eta), "SST": (oc.Ts, oo.Ts)}.items()}
        capped = int(np.sum(np.hypot(oo.uo, oo.vo) > 2.999))
        print(f"ocean step {k}: {errs}, cells at the velocity cap: {capped}, pole rows at the cap: "
              f"{int(np.sum(np.hypot(oo.uo[[0, -1]], oo.vo[[0, -1]]) > 2.999))}")
        assert capped > 1000
        # eta: the sub-step subtracts the area-weighted global mean, a sum over 750 000 cells whose terms reach 1e3 m next to the
        # fast pole rows before the clip -- summation order (fixed-point slots here, pairwise in NumPy) shows at ~1e-12 of the clip
        tol = {"uo": 1e-12, "vo": 1e-12, "eta": 1e-10, "SST": 1e-12}
        for kk, e in errs.items():
            assert e < tol[kk], (k, kk, e)
        for rows in (slice(0, 9), slice(nlat - 9, nlat)):
            for kk, (a, b) in {"uo": (oc.uo, oo.uo), "vo": (oc.vo, oo.vo), "eta": (oc.eta, oo.eta), "SST": (oc.Ts, oo.Ts)}.items():
                assert relerr(a[rows], b[rows]) < tol[kk], (k, kk, rows)


@pytest.mark.parametrize("use_ocean,ka", [
    (1, {"u": -234004.94882386696, "v": 28.279164765879422, "h": 224666201.89422357, "T_s": 8364474.288207765,
         "q": 154.87061864197833, "cloud_cover": 9024.394981981939, "E_flux_last": 1.3207955778646714,
         "W_land": 12872405247.524658, "uo": -27.408644099087475, "vo": -2.4226231764406703,
         "eta": 8.601978767987557, "Ts": 8364361.425071081}),
    (0, {"u": -234004.83009789028, "v": 28.038551435024658, "h": 224666226.2721218, "T_s": 8365678.517250543,
         "q": 154.84011427916005, "cloud_cover": 9024.591677566821, "E_flux_last": 1.3163780338066378,
         "W_land": 12848601558.929781}),
])
def test_device_driver_reproduces_reference_known_answers_a5_a6(gpu, use_ocean, ka):
    """SURVEY.md Appendix A5 / A6 hold the sums of the reference's REAL driver run (121 x 240, seed-42 planet, 24 steps, ecology /
    phytoplankton / routing off).  The device driver must land on them: per-field bound = what DESIGN.md's agreement-horizon table
    shows two f64 implementations keep after 24 coupled steps, expressed on the sum (|dev - ref| <= tol * N * max|field|)."""
    import qingdai_amd as qa
    from qingdai_amd.driver import Simulation
    sim = Simulation(121, 240, params=qa.QdParams(), use_ocean=bool(use_ocean), quiet=True, ecology=False)
    assert int((sim.land_mask == 1).sum()) == 7288
    sim.run_steps(24)
    got = {"u": sim.gcm.u, "v": sim.gcm.v, "h": sim.gcm.h, "T_s": sim.gcm.T_s, "q": sim.gcm.q, "cloud_cover": sim.gcm.cloud_cover,
           "E_flux_last": sim.gcm.E_flux_last, "W_land": sim.dev.get("W_LAND")}
    if use_ocean:
        got.update(uo=sim.ocean.uo, vo=sim.ocean.vo, eta=sim.ocean.eta, Ts=sim.ocean.Ts)
    tol = {"u": 1e-9, "v": 1e-9, "h": 1e-9, "T_s": 1e-9, "q": 1e-9, "E_flux_last": 1e-9, "W_land": 1e-9, "Ts": 1e-9,
           "cloud_cover": 1e-6, "uo": 1e-5, "vo": 1e-5, "eta": 1e-6}
    devs = {}
    for k, s in ka.items():
        a = np.asarray(got[k], dtype=float)
        devs[k] = abs(float(np.sum(a)) - s) / (a.size * max(float(np.max(np.abs(a))), 1e-300))
    print(devs)
    for k, e in devs.items():
        assert e < tol[k], (k, e, float(np.sum(got[k])), ka[k])


def test_config3_eight_bands_match_the_whole_globe_at_1441x2880(gpu):
    """BASELINE configs[3]: 1441 x 2880, full physics, 8 latitude bands (in-process transport on one device) against the
    whole-globe handle: bit-identical atmosphere, ocean to the rounding of the band-wise eta sum."""
    from test_gpu_bands import _run
    nlat, nlon = 1441, 2880
    ref, _ = _run(1, nlat, nlon, 2, dict(energy_w=1.0), True, True)
    got, ex = _run(8, nlat, nlon, 2, dict(energy_w=1.0), True, True)
    print("halo exchanges per band:", ex)
    for k in ("U", "V", "H", "TS", "Q", "CLOUD"):
        assert np.array_equal(got[k], ref[k]), (k, relerr(got[k], ref[k]))
    for k in ("UO", "VO", "ETA", "SST"):
        assert relerr(got[k], ref[k]) < 1e-12, (k, relerr(got[k], ref[k]))


def test_fast_tail_waves_equal_the_general_ones_at_721x1440(gpu, monkeypatch):
    """k_ocn_tail_fast at the benchmark's size and sub-step count: two coupled steps (26 launches, 74 strips x 24 column groups,
    the two edge column groups on ds_bpermute, the pole strips on the general waves) with the slim waves against the same kernel
    with every wave on the general form (QD_TAIL_GENERAL=1: memory gather, clamps, pole stencils).  Same strip cut, hence the same
    order of the eta sum: every ocean field and what the SST write-back does to the atmosphere must agree bit for bit."""
    import bench

    def run(general):
        monkeypatch.setenv("QD_TAIL_GENERAL", general)
        grid, m, oc, forcing, mask, base_albedo, friction = bench.build_case(721, 1440, True)
        import qingdai_amd as qa
        lat = np.deg2rad(grid.lat_mesh); lon = np.deg2rad(grid.lon_mesh)
        m.u, m.v = atsize_wind(lat, lon)
        m._dev.step_n(forcing.star_table([0.0, 300.0]), 300.0, with_ocean=True, with_physics=True, pass_albedo=True)
        assert m._dev.last_ocean_nsub() >= 12
        out = {k: np.array(m._dev.get(k)) for k in ("UO", "VO", "ETA", "SST", "TS", "U", "V", "H", "Q")}
        m._dev.close()
        return out
    fast, gen = run("0"), run("1")
    for k in fast:
        assert np.array_equal(fast[k], gen[k], equal_nan=True), (k, relerr(fast[k], gen[k]))
